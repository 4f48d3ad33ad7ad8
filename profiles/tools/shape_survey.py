#!/usr/bin/env python3
"""Throughput of batches of equal items over item sizes and address alignments (96 MiB a batch, device resident):
looks for shapes that fall off the fast roads.  usage: shape_survey.py [size ...]"""
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
total = 96 << 20
data = harness.splitmix64_bytes(9, total + 64)
d_in = eng.alloc(total + 64)
eng.upload(d_in, data)
sizes = [int(a) for a in sys.argv[1:]] or [3000, 5000, 17000, 33000, 70000, 300000, 1 << 20, 8 << 20]
for size in sizes:
    for odd in (0, 1):
        items = total // (size + odd)
        pitch_in = size + odd                      # odd: every item but the first at an odd address
        cap = (size * 10 // 8 + 64) | odd          # odd: encoded streams at odd addresses too
        d_enc, d_back = eng.alloc(items * cap + 64), eng.alloc(items * size + 64)
        ep = eng.encode_plan([dict(in_offset=odd + i * pitch_in, in_len=size, out_offset=odd + i * cap, out_capacity=cap - 1)
                              for i in range(items)])
        eng.encode_launch(ep, d_in, d_enc)
        res = eng.encode_results(ep, items)
        assert all(r[0] == 0 for r in res)
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(3):
            eng.encode_launch(ep, d_in, d_enc)
        eng.sync()
        t_enc = (time.perf_counter() - t0) / 3
        dp = eng.decode_plan([dict(in_offset=odd + i * cap, in_len=res[i][3], out_offset=i * size, out_capacity=size) for i in range(items)])
        eng.decode_launch(dp, d_enc, d_back)
        dres = eng.decode_results(dp, items)
        assert all(r[0] == 0 and r[2] == size for r in dres)
        got = eng.download(d_back, items * size)
        want = np.concatenate([data[odd + i * pitch_in: odd + i * pitch_in + size] for i in range(min(items, 4))])
        assert np.array_equal(got[:want.size], want)
        t0 = time.perf_counter()
        for _ in range(3):
            eng.decode_launch(dp, d_enc, d_back)
        eng.sync()
        t_dec = (time.perf_counter() - t0) / 3
        n = items * size
        print("%7d items of %8d bytes, %s addresses: encode %7.1f GiB/s, decode %7.1f GiB/s of symbols" % (
            items, size, "odd " if odd else "even", n / 2**30 / t_enc, n / 2**30 / t_dec), flush=True)
        eng.lib.aws_huffman_amd_encode_plan_destroy(ep)
        eng.lib.aws_huffman_amd_decode_plan_destroy(dp)
        eng.free(d_enc)
        eng.free(d_back)
