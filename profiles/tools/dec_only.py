#!/usr/bin/env python3
"""One item of 1 GiB (bench.py's stream): one encode, then decode launches only -- for a kernel timeline under rocprofv3
(profiles/tools/decode_timeline.sh).   usage: dec_only.py [launches]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "tests"))
import harness  # noqa: E402

lib = harness.load_product()
patterns, lens = harness.load_table()
coder = lib.aws_huffman_amd_table_coder_new(patterns, lens)
eng = harness.Engine(lib, coder)
n = int(os.environ.get("DEC_ONLY_MIB", "1024")) << 20
data = harness.splitmix64_bytes(5, n)
cap = n * 10 // 8 + 64
d_in, d_enc, d_back = eng.alloc(n), eng.alloc(cap), eng.alloc(n + 64)
eng.upload(d_in, data)
ep = eng.encode_plan([dict(in_offset=0, in_len=n, out_offset=0, out_capacity=cap)])
eng.encode_launch(ep, d_in, d_enc)
produced = eng.encode_results(ep, 1)[0][3]
dp = eng.decode_plan([dict(in_offset=0, in_len=produced, out_offset=0, out_capacity=n)])
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    eng.decode_launch(dp, d_enc, d_back)
eng.sync()
res = eng.decode_results(dp, 1)
assert res[0][0] == 0 and res[0][2] == n, res
print("done")
