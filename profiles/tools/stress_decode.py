#!/usr/bin/env python3
"""Soak for the decode launches that use the engine's second stream (>= 1024 chunks): one long stream and a batch of
16 KiB streams, decoded again and again, every output compared with the input; prints the slowest launches and any
difference."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "tests"))
import harness  # noqa: E402

lib = harness.load_product()
patterns, lens = harness.load_table()
coder = lib.aws_huffman_amd_table_coder_new(patterns, lens)
eng = harness.Engine(lib, coder)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 120
total = 96 << 20
data = harness.splitmix64_bytes(9, total)
d_in = eng.alloc(total)
eng.upload(d_in, data)
for size in (total, 16384, 70000):
    items = total // size
    cap = size * 10 // 8 + 64
    d_enc, d_back = eng.alloc(items * cap), eng.alloc(items * size + 64)
    ep = eng.encode_plan([dict(in_offset=i * size, in_len=size, out_offset=i * cap, out_capacity=cap) for i in range(items)])
    eng.encode_launch(ep, d_in, d_enc)
    res = eng.encode_results(ep, items)
    assert all(r[0] == 0 for r in res)
    dp = eng.decode_plan([dict(in_offset=i * cap, in_len=res[i][3], out_offset=i * size, out_capacity=size) for i in range(items)])
    times, bad = [], 0
    for k in range(rounds):
        eng.fill(d_back, 0x5A, items * size + 64)
        eng.sync()
        t0 = time.perf_counter()
        eng.decode_launch(dp, d_enc, d_back)
        eng.sync()
        times.append(time.perf_counter() - t0)
        dres = eng.decode_results(dp, items)
        got = eng.download(d_back, items * size)
        if not all(r[0] == 0 and r[2] == size for r in dres) or not np.array_equal(got, data[:items * size]):
            bad += 1
    t = np.array(times) * 1e3
    print("%6d streams of %9d bytes: %d decode launches, median %.3f ms, max %.3f ms, wrong: %d" % (
        items, size, rounds, np.median(t), t.max(), bad), flush=True)
    eng.lib.aws_huffman_amd_decode_plan_destroy(dp)
    eng.lib.aws_huffman_amd_encode_plan_destroy(ep)
    eng.free(d_enc)
    eng.free(d_back)
