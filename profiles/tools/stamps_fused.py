#!/usr/bin/env python3
"""Where does a workgroup of dec_onepass spend its cycles?  (diagnostic build only: `make -C aws-c-compression_amd stamps`)

Shader clocks between the in-kernel stamps of wave 0, averaged over the chunks of a 1 GiB stream.  Shares only; never
quote this build's run time."""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "tests"))
import harness  # noqa: E402

os.environ.setdefault("AWS_HUFFMAN_AMD_DECODE", "one-pass")
import numpy as np  # noqa: E402

lib = harness.load_product(os.path.join(REPO, "aws-c-compression_amd", "libaws-c-compression-amd-stamps.so"))
lib.hufk_stamps_attach.argtypes = [C.c_void_p]
patterns, lens = harness.load_table()
eng = harness.Engine(lib, lib.aws_huffman_amd_table_coder_new(patterns, lens))
MAX_WG = 131072
d_rows = eng.alloc(3 * MAX_WG * 8 * 8)
assert lib.hufk_stamps_attach(d_rows) == 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
d_in, d_enc, d_back = eng.alloc(n), eng.alloc(n * 10 // 8 + 64), eng.alloc(n + 64)
eng.fill_splitmix64(d_in, n, 5)
ep = eng.encode_plan([dict(in_offset=0, in_len=n, out_offset=0, out_capacity=n * 10 // 8 + 64)])
eng.encode_launch(ep, d_in, d_enc)
e_len = eng.encode_results(ep, 1)[0][3]
dp = eng.decode_plan([dict(in_offset=0, in_len=e_len, out_offset=0, out_capacity=n)])
eng.upload(d_rows, np.zeros(3 * MAX_WG * 8, dtype=np.uint64).view(np.uint8))
for _ in range(2):
    eng.decode_launch(dp, d_enc, d_back)
    eng.decode_results(dp, 1)
print("road", eng.decode_road(dp), eng.last_road_detail)
r = eng.download(d_rows, 4096 * 64, offset=0).view(np.uint64).reshape(4096, 8).astype(np.float64)
r = r[r[:, 7] > 0]  # persistent kernel: one row per resident workgroup, sums over the tiles of its wave 0 (both launches)
tiles = r[:, 7]
d = np.diff(r[:, :7], axis=1)
per_tile = d / tiles[:, None]
life = per_tile.sum(axis=1)
print("dec_onepass: %d resident workgroups, %.1f tiles per wave; %.0f clocks per tile (median %.0f), wave 0 of each" % (
    len(r), tiles.mean() / 2, life.mean(), np.median(life)))
names = ["R: the walk from the meeting bit, symbols to the slot", "H: the first rows again from the true entry state",
         "counts, count published", "the NEXT tile's record, rows and U (all entry states to one head)",
         "wait for the offsets in front", "slots to HBM"]
for i, ph in enumerate(names):
    print("   %-72s %9.0f  %5.1f %%" % (ph, per_tile[:, i].mean(), 100 * per_tile[:, i].mean() / life.mean()))
compute = per_tile[:, [0, 1, 2, 3, 5]].sum(axis=1)
wait = per_tile[:, 4]
q = [0, 1, 10, 50, 90, 99, 100]
print("compute clocks per tile over the workgroups, percentiles %s: %s" % (q, [int(x) for x in np.percentile(compute, q)]))
print("wait    clocks per tile over the workgroups, percentiles %s: %s" % (q, [int(x) for x in np.percentile(wait, q)]))
idx = np.flatnonzero(r0_mask) if 'r0_mask' in globals() else np.arange(len(compute))
for lo in range(0, len(compute), 64):
    print("  workgroups %3d..%3d: compute %6.0f wait %6.0f" % (lo, min(lo + 63, len(compute) - 1), compute[lo:lo + 64].mean(), wait[lo:lo + 64].mean()))
