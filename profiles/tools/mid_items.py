#!/usr/bin/env python3
"""Batches of items between header size and a whole segment/chunk (0.6 .. 12 KiB each, 128 MiB a batch), device
resident, one plan per size.  For DESIGN.md.   usage: mid_items.py [size ...]"""
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
oracle = harness.oracle_codec()
ocoder = oracle.lib.oracle_table_coder_new(patterns, lens)
eng = harness.Engine(lib, coder)
total = int(os.environ.get("MID_ITEMS_TOTAL_MIB", "128")) << 20  # (a smaller batch has fewer items: a class may then leave the thread-per-item road)
data = harness.splitmix64_bytes(6, total)
d_in, d_back = eng.alloc(total), eng.alloc(total + 64)
eng.upload(d_in, data)
for size in [int(a) for a in sys.argv[1:]] or [600, 1024, 2048, 4096, 8192, 12288]:
    items = total // size
    cap = size * 10 // 8 + 8
    d_enc = eng.alloc(items * cap)
    ep = eng.encode_plan([dict(in_offset=i * size, in_len=size, out_offset=i * cap, out_capacity=cap) for i in range(items)])
    eng.encode_launch(ep, d_in, d_enc)
    res = eng.encode_results(ep, items)
    assert all(r[0] == 0 for r in res)
    t0 = time.perf_counter()
    for _ in range(3):
        eng.encode_launch(ep, d_in, d_enc)
    eng.sync()
    t_enc = (time.perf_counter() - t0) / 3
    dp = eng.decode_plan([dict(in_offset=i * cap, in_len=res[i][3], out_offset=i * size, out_capacity=size) for i in range(items)])
    eng.decode_launch(dp, d_enc, d_back)
    dres = eng.decode_results(dp, items)
    assert all(r[0] == 0 and r[2] == size for r in dres)
    assert np.array_equal(eng.download(d_back, items * size), data[:items * size])
    for i in (0, items // 2, items - 1):
        want = oracle.encode_all(ocoder, data[i * size:(i + 1) * size])
        assert np.array_equal(eng.download(d_enc, res[i][3], offset=i * cap), want), i
    t0 = time.perf_counter()
    for _ in range(3):
        eng.decode_launch(dp, d_enc, d_back)
    eng.sync()
    t_dec = (time.perf_counter() - t0) / 3
    print("%6d items of %5d bytes: encode %7.2f ms = %6.1f GiB/s; decode %7.2f ms = %6.1f GiB/s of symbols" % (
        items, size, t_enc * 1e3, total / 2**30 / t_enc, t_dec * 1e3, total / 2**30 / t_dec), flush=True)
    eng.lib.aws_huffman_amd_encode_plan_destroy(ep)
    eng.lib.aws_huffman_amd_decode_plan_destroy(dp)
    eng.free(d_enc)
