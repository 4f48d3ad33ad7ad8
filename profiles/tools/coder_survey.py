#!/usr/bin/env python3
"""Encode / decode rate of ONE 128 MiB stream for the coder profiles of tests/parity_cases.py (code-length shapes other
than the test coder's), symbols drawn to match the code lengths and uniformly: which roads they take and how fast."""
import ctypes as C
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "tests"))
import harness  # noqa: E402
import parity_cases as pc  # noqa: E402

lib = harness.load_product()
n = 128 << 20
eng = None
slowest_at = 0


def timed(launch, reps):
    """median seconds of one launch + wait, each launch timed by itself, and the slowest one's ratio to the median"""
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        launch()
        eng.sync()
        times.append(time.perf_counter() - t0)
    global slowest_at
    slowest_at = times.index(max(times))
    times.sort()
    return times[len(times) // 2], times[-1] / times[len(times) // 2]


rng = np.random.default_rng(3)
for name, rows in pc.CODER_PROFILES.items():
    lengths = [l for count, l in rows for _ in range(count)]
    patterns, lens = pc.canonical_code(lengths)
    coder = lib.aws_huffman_amd_table_coder_new((C.c_uint32 * 256)(*patterns), (C.c_uint8 * 256)(*lens))
    eng = harness.Engine(lib, coder)
    prob = np.array([2.0 ** -l for l in lengths])
    prob /= prob.sum()
    for kind in ("matched", "uniform"):
        data = (rng.choice(256, size=n, p=prob) if kind == "matched" else rng.integers(0, 256, n)).astype(np.uint8)
        bits = int(np.array(lengths, dtype=np.int64)[data].sum())
        cap = (bits + 7) // 8 + 64
        d_in, d_enc, d_back = eng.alloc(n), eng.alloc(cap), eng.alloc(n + 64)
        eng.upload(d_in, data)
        ep = eng.encode_plan([dict(in_offset=0, in_len=n, out_offset=0, out_capacity=cap)])
        eng.encode_launch(ep, d_in, d_enc)
        (rc, err, consumed, e_len, _, _), = eng.encode_results(ep, 1)
        assert rc == 0 and e_len == (bits + 7) // 8, (name, rc, err, e_len, bits)
        eng.sync()
        t_enc, enc_worst = timed(lambda: eng.encode_launch(ep, d_in, d_enc), 7)
        enc_at = slowest_at
        one_pass = bool(lib.aws_huffman_amd_engine_encodes_in_one_pass(eng.h))
        line = "%-14s %-8s %5.2f bits/symbol: encode %7.1f GiB/s (%s)" % (
            name, kind, bits / n, n / 2**30 / t_enc, "one pass" if one_pass else "three kernels")
        if True:
            dp = eng.decode_plan([dict(in_offset=0, in_len=e_len, out_offset=0, out_capacity=n)])
            eng.decode_launch(dp, d_enc, d_back)
            (rc, err, symbols, _), = eng.decode_results(dp, 1)
            assert rc == 0 and symbols == n, (name, rc, err, symbols)
            assert np.array_equal(eng.download(d_back, n), data)
            t_dec, dec_worst = timed(lambda: eng.decode_launch(dp, d_enc, d_back), 7)
            line += ", decode %7.1f GiB/s of symbols" % (n / 2**30 / t_dec)
            if max(enc_worst, dec_worst) > 1.5:
                # (round 2's table had one such line, taken for a stall of the kernels: it is the host's clock around
                # three launches that was timed then)
                line += " [slowest of 7 launches / median: encode %.2f (launch %d), decode %.2f (launch %d)]" % (
                    enc_worst, enc_at, dec_worst, slowest_at)
            eng.lib.aws_huffman_amd_decode_plan_destroy(dp)
        print(line, flush=True)
        eng.lib.aws_huffman_amd_encode_plan_destroy(ep)
        for p in (d_in, d_enc, d_back):
            eng.free(p)
