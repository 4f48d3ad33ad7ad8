#!/usr/bin/env python3
"""Where does a workgroup of the decode kernels spend its cycles?  (diagnostic build only)

Runs encode + decode of a stream through libaws-c-compression-amd-stamps.so (built with
`make -C aws-c-compression_amd stamps`, -DHUFD_STAMPS) and prints the average number of shader
clocks between the in-kernel stamps.  Shares only; never quote this build's run time.
"""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "tests"))
import harness  # noqa: E402

lib = harness.load_product(os.path.join(REPO, "aws-c-compression_amd", "libaws-c-compression-amd-stamps.so"))
lib.hufk_stamps_attach.argtypes = [C.c_void_p]
patterns, lens = harness.load_table()
eng = harness.Engine(lib, lib.aws_huffman_amd_table_coder_new(patterns, lens))
import numpy as np

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


def rows(kernel, n_wg, n_ph):
    raw = eng.download(d_rows, n_wg * 64, offset=kernel * MAX_WG * 64).view(np.uint64).reshape(n_wg, 8)
    return raw[:, :n_ph].astype(np.float64)


def report(name, r, phases):
    d = np.diff(r, axis=1)
    life = r[:, -1] - r[:, 0]
    print("%s: %.0f clocks per workgroup (median %.0f), wave 0" % (name, life.mean(), np.median(life)))
    for i, ph in enumerate(phases):
        print("   %-44s %9.0f  %5.1f %%" % (ph, d[:, i].mean(), 100 * d[:, i].mean() / life.mean()))


def zero_rows():
    z = np.zeros(3 * MAX_WG * 8, dtype=np.uint64)
    eng.upload(d_rows, z.view(np.uint8))


zero_rows()
eng.encode_launch(ep, d_in, d_enc)
eng.encode_results(ep, 1)
r = rows(2, 4096, 8)
r = r[r[:, 0] > 0]  # persistent kernel: one row per resident workgroup, sums over its tiles
tiles = (n + 4095) // 4096
per = tiles / 8.0 / max(len(r), 1)  # iterations of a workgroup's wave 0
print("enc_onepass: %d resident workgroups, %.1f tiles per wave" % (len(r), per))
rr = np.stack([r[:, 0], r[:, 1], r[:, 7], r[:, 2], r[:, 3], r[:, 4], r[:, 5]], axis=1)
report("enc_onepass (sum over the tiles of wave 0 of each workgroup)", rr,
       ["fresh tile: descriptors, loads, look-ups, octs, scan, publish", "old tile: wait for the offsets asked for at the top of the turn",
        "old tile: polls (offsets not there yet)", "old tile: records, last byte, shifted copy-out", "fresh tile: octs into the image", "-"])
print("   polls per tile: %.2f" % (r[:, 6].mean() / per))
for _ in range(2):
    eng.decode_launch(dp, d_enc, d_back)
    eng.decode_results(dp, 1)
chunks = (e_len + 32767) // 32768
r = rows(0, min(chunks, MAX_WG), 8)[1:]  # (row 0 is also the end-of-stream instantiation's)
r = np.stack([r[:, 0], r[:, 1], r[:, 2], r[:, 3], r[:, 4], r[:, 7], r[:, 5], r[:, 6]], axis=1)
r = r[(np.diff(r, axis=1) >= 0).all(axis=1) & (r[:, 0] > 0)]  # (the chunk the stream ends in is the other instantiation's: no stamps)
report("dec_sync_one", r,
       ["loads, table, barrier", "-", "phase R (one guessed walk over the sub-chunk)", "wait barrier",
        "phase H (own entry to where it meets the guessed walk)", "sub-chunk 0's candidates (wave 0)", "sums, barrier, records out"])
report("dec_emit", rows(1, min(chunks, MAX_WG), 6), ["load", "entry chains", "walk", "wait barrier", "copy out"])
