#!/usr/bin/env python3
"""What the encoder's way back costs when it is taken: a 1 GiB stream whose one-pass kernel is made to give up half-way
(aws_huffman_amd_testing_set_encode_road(..ONE_PASS_FAILS)), the launch timed with events and its output digested; beside it
the launch that does not give up, and the three-kernel road.   usage: way_back_time.py [MiB]"""
import hashlib
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "tests"))
import harness  # noqa: E402

lib = harness.load_product()
patterns, lens = harness.load_table()
coder = lib.aws_huffman_amd_table_coder_new(patterns, lens)
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024) << 20
cap = n * 10 // 8 + 64
digests = {}
for road in (None, "one-pass-fails", "three-kernel"):
    with harness.encode_road(lib, road):
        eng = harness.Engine(lib, coder)
    d_in, d_out = eng.alloc(n), eng.alloc(cap)
    eng.fill_splitmix64(d_in, n, 5)
    plan = eng.encode_plan([dict(in_offset=0, in_len=n, out_offset=0, out_capacity=cap)])
    ev = eng.new_events(2)
    times = []
    for k in range(5):
        eng.fill(d_out, 0xA5, cap)
        eng.record(ev[0])
        eng.encode_launch(plan, d_in, d_out)
        eng.record(ev[1])
        eng.sync()
        times.append(eng.elapsed_ms(ev[0], ev[1]))
        if k == 0:
            res = eng.encode_results(plan, 1)[0]
            digests[road] = (res, hashlib.sha256(eng.download(d_out, res[3]).tobytes()).hexdigest())
            road_taken = eng.encode_road(plan)
            # (the fetch tells the plan that its one-pass kernel gave up: it keeps to the three kernels from then on --
            #  a fresh plan for the launches that are timed)
            lib.aws_huffman_amd_encode_plan_destroy(plan)
            plan = eng.encode_plan([dict(in_offset=0, in_len=n, out_offset=0, out_capacity=cap)])
    print("%-16s road %d  launches %s ms  %s" % (road, road_taken, " ".join("%.3f" % t for t in times), digests[road][1][:16]), flush=True)
    eng.free(d_in)
    eng.free(d_out)
    eng.close()
assert len({v for v in digests.values()}) == 1, digests
print("the three roads wrote the same bytes and records")
