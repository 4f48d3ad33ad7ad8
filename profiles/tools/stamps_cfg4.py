#!/usr/bin/env python3
"""Where does a workgroup of dec_emit_fast spend its clocks on BASELINE config 4 (65 536 streams of 16 KiB, every chunk
holds the end of its stream)?  Diagnostic build only (make -C aws-c-compression_amd stamps); shares, never run times."""
import ctypes as C
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "tests"))
import harness  # noqa: E402

lib = harness.load_product(os.path.join(REPO, "aws-c-compression_amd", "libaws-c-compression-amd-stamps.so"))
lib.hufk_stamps_attach.argtypes = [C.c_void_p]
patterns, lens = harness.load_table()
eng = harness.Engine(lib, lib.aws_huffman_amd_table_coder_new(patterns, lens))
MAX_WG = 131072
d_rows = eng.alloc(3 * MAX_WG * 8 * 8)
assert lib.hufk_stamps_attach(d_rows) == 0
count, size = 65536, 16384
stride = 2 * size
d_in, d_out, d_back = eng.alloc(count * size), eng.alloc(count * stride), eng.alloc(count * size)
eng.fill_splitmix64(d_in, count * size, 7)
ep = eng.encode_plan([dict(in_offset=k * size, in_len=size, out_offset=k * stride, out_capacity=stride) for k in range(count)])
eng.encode_launch(ep, d_in, d_out)
res = eng.encode_results(ep, count)
dp = eng.decode_plan([dict(in_offset=k * stride, in_len=res[k][3], out_offset=k * size, out_capacity=size) for k in range(count)])
eng.upload(d_rows, np.zeros(3 * MAX_WG * 64, dtype=np.uint8))
for _ in range(2):
    eng.decode_launch(dp, d_out, d_back)
    eng.decode_results(dp, count)
raw = eng.download(d_rows, count * 64, offset=1 * MAX_WG * 64).view(np.uint64).reshape(count, 8)[:, :6].astype(np.float64)
raw = raw[raw[:, 0] > 0]
d = np.diff(raw, axis=1)
life = raw[:, -1] - raw[:, 0]
print("dec_emit_fast<TAIL>: %d workgroups, %.0f clocks each (median %.0f); span of the launch %.0f clocks" % (
    len(raw), life.mean(), np.median(life), raw[:, -1].max() - raw[:, 0].min()))
for i, ph in enumerate(["descriptor + loads + table", "scan of the lane counts (two barriers)", "chain set-up", "walk", "wait barrier", "copy out"][1:]):
    print("   %-44s %9.0f  %5.1f %%" % (ph, d[:, i].mean(), 100 * d[:, i].mean() / life.mean()))
print("   first stamp after the launch's first: mean %.0f" % (raw[:, 0] - raw[:, 0].min()).mean())
