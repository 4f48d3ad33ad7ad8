#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in entry points on HOST buffers (aws_huffman_encode /
aws_huffman_decode with host pointers: H2D + kernels + D2H), and the batched device-pointer API
on BASELINE config 4 (65 536 x 16 KiB buffers).  Reported in DESIGN.md, never as bench.py's value."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "tests"))
import harness  # noqa: E402

lib = harness.load_product()
codec = harness.Codec(lib, "aws_")
patterns, lens = harness.load_table()
coder = lib.aws_huffman_amd_table_coder_new(patterns, lens)
GIB = float(1 << 30)

n = 256 << 20
data = harness.splitmix64_bytes(5, n)
enc_buf = np.ones(n * 5 // 4 + 4096, np.uint8)   # touched: no first-use page faults inside the timed calls
back = np.ones(n, np.uint8)
for rep in range(3):
    e = codec.new_encoder(coder)
    t0 = time.perf_counter()
    r = codec.encode_call(e, data, 0, enc_buf, 0, enc_buf.size)
    t1 = time.perf_counter()
    assert r.rc == 0 and r.consumed == n
    d = codec.new_decoder(coder)
    t2 = time.perf_counter()
    r2 = codec.decode_call(d, enc_buf, 0, r.produced, back, 0, n)
    t3 = time.perf_counter()
    assert r2.rc == 0 and r2.produced == n
assert np.array_equal(back, data)
print("host-pointer C ABI (one aws_huffman_encode / aws_huffman_decode call on host memory), %d MiB: encode %.1f ms = %.2f GiB/s of input, decode %.1f ms = %.2f GiB/s of symbols" % (
    n >> 20, (t1 - t0) * 1e3, n / GIB / (t1 - t0), (t3 - t2) * 1e3, n / GIB / (t3 - t2)))

# config 4: 65 536 independent 16 KiB buffers, device resident, one plan
eng = harness.Engine(lib, coder)
items, item_len = 65536, 16384
total = items * item_len
d_in = eng.alloc(total)
eng.fill_splitmix64(d_in, total, 2)
cap = 2 * item_len
d_enc = eng.alloc(items * cap)
plan = eng.encode_plan([dict(in_offset=i * item_len, in_len=item_len, out_offset=i * cap, out_capacity=cap) for i in range(items)])
eng.encode_launch(plan, d_in, d_enc)
res = eng.encode_results(plan, items)
assert all(r[0] == 0 for r in res)
t0 = time.perf_counter()
for _ in range(5):
    eng.encode_launch(plan, d_in, d_enc)
eng.sync()
t1 = time.perf_counter()
d_back = eng.alloc(total + 64)
dplan = eng.decode_plan([dict(in_offset=i * cap, in_len=res[i][3], out_offset=i * item_len, out_capacity=item_len) for i in range(items)])
eng.decode_launch(dplan, d_enc, d_back)
dres = eng.decode_results(dplan, items)
assert all(r[0] == 0 and r[2] == item_len for r in dres)
t2 = time.perf_counter()
for _ in range(5):
    eng.decode_launch(dplan, d_enc, d_back)
eng.sync()
t3 = time.perf_counter()
print("config 4 (65 536 x 16 KiB, device resident): encode %.2f ms = %.0f GiB/s, decode %.2f ms = %.0f GiB/s of symbols" % (
    (t1 - t0) / 5 * 1e3, total / GIB / ((t1 - t0) / 5), (t3 - t2) / 5 * 1e3, total / GIB / ((t3 - t2) / 5)))
