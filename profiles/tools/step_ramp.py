#!/usr/bin/env python3
"""GPU box: how long the bench's step takes as a function of how long the GPU has been busy.  After an idle stretch
(seconds: what bench.py's set-up leaves behind -- its digests are taken on the host) the 1 GiB step is launched N times
back to back, an event between steps, and every step's duration is printed: the first steps run on clocks that are still
coming up.  Not part of the graded runs; it says how much of `ms_per_step` at small --steps is the ramp.

  usage: step_ramp.py [steps=150] [idle_seconds=2] [bytes=1 GiB]
"""
import os
import statistics
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "tests"))
import harness  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
idle = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 30
lib = harness.load_product()
patterns, lens = harness.load_table()
eng = harness.Engine(lib, lib.aws_huffman_amd_table_coder_new(patterns, lens))
worst = n * 10 // 8 + 64
d_in, d_enc, d_back = eng.alloc(n), eng.alloc(worst), eng.alloc(n + 64)
eng.fill_splitmix64(d_in, n, 5)
enc_plan = eng.encode_plan([dict(in_offset=0, in_len=n, out_offset=0, out_capacity=worst)])
eng.encode_launch(enc_plan, d_in, d_enc)
(rc, err, consumed, e_len, _, _), = eng.encode_results(enc_plan, 1)
dec_plan = eng.decode_plan([dict(in_offset=0, in_len=e_len, out_offset=0, out_capacity=n)])
eng.decode_launch(dec_plan, d_enc, d_back)
assert eng.decode_results(dec_plan, 1)[0][2] == n
for rep in range(2):
    ev = eng.new_events(steps + 1)
    eng.sync()
    time.sleep(idle)
    t0 = time.perf_counter()
    eng.record(ev[0])
    for k in range(steps):
        eng.encode_launch(enc_plan, d_in, d_enc)
        eng.decode_launch(dec_plan, d_enc, d_back)
        eng.record(ev[k + 1])
    eng.sync()
    wall = time.perf_counter() - t0
    ms = [eng.elapsed_ms(ev[k], ev[k + 1]) for k in range(steps)]
    print("after %.1f s idle: %d steps in %.2f ms of wall clock" % (idle, steps, wall * 1e3))
    for a in range(0, steps, 10):
        print("  steps %3d..%3d  " % (a, min(a + 10, steps) - 1) + " ".join("%.3f" % x for x in ms[a:a + 10]))
    tail = statistics.median(ms[steps // 2:])
    print("  median of the second half %.4f ms; the first 5 / 10 / 25 steps cost %.3f / %.3f / %.3f ms more than that many at the median" % (
        tail, sum(ms[:5]) - 5 * tail, sum(ms[:10]) - 10 * tail, sum(ms[:25]) - 25 * tail))
