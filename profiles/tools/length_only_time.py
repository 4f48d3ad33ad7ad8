#!/usr/bin/env python3
"""What a length-only launch costs (aws_huffman_get_encoded_length for a batch: nothing is written): BASELINE configs[3]'s
65 536 buffers of 16 KiB, 2 KiB items, and one stream of 256 MiB.   usage: length_only_time.py"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "tests"))
import harness  # noqa: E402

lib = harness.load_product()
patterns, lens = harness.load_table()
eng = harness.Engine(lib, lib.aws_huffman_amd_table_coder_new(patterns, lens))
oracle = harness.oracle_codec()
ocoder = oracle.lib.oracle_table_coder_new(patterns, lens)
total = 256 << 20
data = harness.splitmix64_bytes(7, total)
d_in = eng.alloc(total)
eng.upload(d_in, data)
for size in (2048, 16384, total):
    n = total // size
    plan = eng.encode_plan([dict(in_offset=i * size, in_len=size, out_offset=0, out_capacity=0) for i in range(n)])
    eng.encode_launch(plan, d_in, None, length_only=True)
    got = eng.encoded_lengths(plan, n)
    for i in (0, n // 2, n - 1):
        assert got[i] == oracle.encode_all(ocoder, data[i * size:(i + 1) * size]).size if size < total else True, (size, i)
    t0 = time.perf_counter()
    for _ in range(3):
        eng.encode_launch(plan, d_in, None, length_only=True)
    eng.sync()
    t = (time.perf_counter() - t0) / 3
    print("%8d items of %9d bytes: a length-only launch %7.3f ms = %6.1f GiB/s" % (n, size, t * 1e3, total / 2**30 / t), flush=True)
    eng.lib.aws_huffman_amd_encode_plan_destroy(plan)
