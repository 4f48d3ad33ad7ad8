#!/usr/bin/env python3
"""Which road does the decode of an n-byte stream (seed 5) take, and if dec_onepass gives up: which tile, why?"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "tests"))
import harness  # noqa: E402

os.environ.setdefault("AWS_HUFFMAN_AMD_DECODE", "one-pass")

lib = harness.load_product(os.environ.get("HUF_LIB"))
patterns, lens = harness.load_table()
eng = harness.Engine(lib, lib.aws_huffman_amd_table_coder_new(patterns, lens))
for n in [int(x) for x in sys.argv[1:]] or [1 << 24, 1 << 27, 1 << 30]:
    d_in, d_enc, d_back = eng.alloc(n), eng.alloc(n * 10 // 8 + 64), eng.alloc(n + 64)
    eng.fill_splitmix64(d_in, n, 5)
    ep = eng.encode_plan([dict(in_offset=0, in_len=n, out_offset=0, out_capacity=n * 10 // 8 + 64)])
    eng.encode_launch(ep, d_in, d_enc)
    e_len = eng.encode_results(ep, 1)[0][3]
    dp = eng.decode_plan([dict(in_offset=0, in_len=e_len, out_offset=0, out_capacity=n)])
    ev = eng.new_events(2)
    for rep in range(3):
        eng.record(ev[0])
        eng.decode_launch(dp, d_enc, d_back)
        eng.record(ev[1])
        res = eng.decode_results(dp, 1)
        road = eng.decode_road(dp)
        print("n=%d e=%d tiles~%d road %d detail(tile,why,walked twice)=%s ms=%.3f res=%s" % (
            n, e_len, e_len // 8064, road, eng.last_road_detail, eng.elapsed_ms(ev[0], ev[1]), res), flush=True)
    for q in (d_in, d_enc, d_back):
        eng.free(q)
