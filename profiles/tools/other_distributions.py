#!/usr/bin/env python3
"""Throughput on inputs other than uniform random bytes (device resident, one plan each), for DESIGN.md:
printable ASCII, a text-like skew towards the shortest codes, and two streams that do NOT self-synchronise
(one symbol repeated; two 5-bit symbols at random), which take the long way through the decoder."""
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
n = 256 << 20
L = np.array([lens[i] for i in range(256)])
for kind in ("printable", "short-codes", "one-symbol", "two-symbols"):
    raw = harness.splitmix64_bytes(4, n)
    if kind == "printable":
        data = (32 + raw % 95).astype(np.uint8)
    elif kind == "short-codes":
        pool = np.flatnonzero(L <= 6).astype(np.uint8)  # the 19 symbols with codes of 5 and 6 bits
        data = pool[raw % pool.size]
    elif kind == "one-symbol":
        data = np.full(n, ord("e"), np.uint8)
    else:
        data = np.where(raw & 1, ord("a"), ord(" ")).astype(np.uint8)
    d_in, d_enc, d_back = eng.alloc(n), eng.alloc(n * 2 + 64), eng.alloc(n + 64)
    eng.upload(d_in, data)
    ep = eng.encode_plan([dict(in_offset=0, in_len=n, out_offset=0, out_capacity=n * 2)])
    eng.encode_launch(ep, d_in, d_enc)
    (rc, err, consumed, e_len, _, _), = eng.encode_results(ep, 1)
    assert rc == 0, (rc, err)
    dp = eng.decode_plan([dict(in_offset=0, in_len=e_len, out_offset=0, out_capacity=n)])
    eng.decode_launch(dp, d_enc, d_back)
    (rc, err, symbols, _), = eng.decode_results(dp, 1)
    assert rc == 0 and symbols == n, (rc, err, symbols)
    assert np.array_equal(eng.download(d_back, n), data)
    # HIP events around the launches (host clocks around a few launches after a long host-side check pick up
    # whatever the GPU's clocks are doing at that moment)
    ev = eng.new_events(4)
    t_enc = t_dec = 0.0
    for _ in range(5):
        eng.encode_launch(ep, d_in, d_enc, events=ev)
        eng.sync()
        t_enc += eng.elapsed_ms(ev[0], ev[3]) / 5e3
    for _ in range(5):
        eng.decode_launch(dp, d_enc, d_back, events=ev)
        eng.sync()
        t_dec += eng.elapsed_ms(ev[0], ev[3]) / 5e3
    print("%-12s %d MiB -> %.2f bits/symbol: encode %.0f GiB/s, decode %.0f GiB/s of symbols" % (
        kind, n >> 20, e_len * 8 / n, n / 2**30 / t_enc, n / 2**30 / t_dec), flush=True)
    for p in (d_in, d_enc, d_back):
        eng.free(p)
