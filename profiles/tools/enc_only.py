#!/usr/bin/env python3
"""One item of 1 GiB (bench.py's stream), encode launches only, nothing checked: for kernel durations under rocprofv3 of
builds whose output is not meant to be right (an experiment's upper bound).   usage: enc_only.py [launches]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "tests"))
import harness  # noqa: E402

lib = harness.load_product()
patterns, lens = harness.load_table()
coder = lib.aws_huffman_amd_table_coder_new(patterns, lens)
eng = harness.Engine(lib, coder)
n = int(os.environ.get("ENC_ONLY_MIB", "1024")) << 20
data = harness.splitmix64_bytes(5, n)
cap = n * 10 // 8 + 64
d_in, d_out = eng.alloc(n), eng.alloc(cap)
eng.upload(d_in, data)
plan = eng.encode_plan([dict(in_offset=0, in_len=n, out_offset=0, out_capacity=cap)])
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    eng.encode_launch(plan, d_in, d_out)
eng.sync()
print("done")
