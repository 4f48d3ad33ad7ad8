#!/usr/bin/env python3
"""Latency of one aws_huffman_encode / aws_huffman_decode call on a header-sized host string (the reference's
HPACK use: one call per header field).  For DESIGN.md."""
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
for n in (15, 300, 1024, 4096, 8192, 16384, 24576, 65536):
    data = (32 + harness.splitmix64_bytes(3, n) % 95).astype(np.uint8)
    enc = np.zeros(2 * n + 16, np.uint8)
    back = np.zeros(n, np.uint8)
    reps = 300
    for timed in (False, True):
        t_enc = t_dec = 0.0
        for _ in range(reps):
            e = codec.new_encoder(coder)
            t0 = time.perf_counter()
            r = codec.encode_call(e, data, 0, enc, 0, enc.size)
            t1 = time.perf_counter()
            d = codec.new_decoder(coder)
            t2 = time.perf_counter()
            r2 = codec.decode_call(d, enc, 0, r.produced, back, 0, n)
            t3 = time.perf_counter()
            t_enc += t1 - t0
            t_dec += t3 - t2
        assert r.rc == 0 and r2.rc == 0 and np.array_equal(back, data)
    print("%6d bytes: aws_huffman_encode %.1f us a call, aws_huffman_decode %.1f us a call (host pointers, ctypes overhead included)" % (
        n, t_enc / reps * 1e6, t_dec / reps * 1e6), flush=True)
