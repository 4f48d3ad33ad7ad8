#!/usr/bin/env python3
"""Soak for the one-pass encoder's look-back: many launches of plans with many small items, each timed on its own and
its output compared with the first launch's; prints the slowest launches and any difference."""
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
total = 128 << 20
data = harness.splitmix64_bytes(6, total)
d_in = eng.alloc(total)
eng.upload(d_in, data)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 150
for size in (600, 1024, 2048, 3000, 4096, 16384, 1 << 20):
    items = total // size
    cap = size * 10 // 8 + 8
    d_enc = eng.alloc(items * cap)
    ep = eng.encode_plan([dict(in_offset=i * size, in_len=size, out_offset=i * cap, out_capacity=cap) for i in range(items)])
    eng.encode_launch(ep, d_in, d_enc)
    res = eng.encode_results(ep, items)
    assert all(r[0] == 0 for r in res)
    first = eng.download(d_enc, items * cap).copy()
    times, bad = [], 0
    for k in range(rounds):
        eng.fill(d_enc, 0, items * cap) if k % 10 == 0 else None
        eng.sync()
        t0 = time.perf_counter()
        eng.encode_launch(ep, d_in, d_enc)
        eng.sync()
        times.append(time.perf_counter() - t0)
        if k % 10 == 0:
            r2 = eng.encode_results(ep, items)
            got = eng.download(d_enc, items * cap)
            if r2 != res or not np.array_equal(got, first):
                bad += 1
    t = np.array(times) * 1e3
    print("%8d items of %7d bytes: %d launches, median %.3f ms, max %.3f ms, over 3x median: %d, wrong: %d" % (
        items, size, rounds, np.median(t), t.max(), int((t > 3 * np.median(t)).sum()), bad), flush=True)
    eng.lib.aws_huffman_amd_encode_plan_destroy(ep)
    eng.free(d_enc)
