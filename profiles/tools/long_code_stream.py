#!/usr/bin/env python3
"""One long stream of a coder with codes of more than 12 bits (tests/parity_cases.py CODER_PROFILES, default
hpack_lengths): encode through the segment kernels, decode a workgroup per 32 KiB block (dec_wide_*).  `len4to15` gives
the printable symbols codes of 9, 12 and 15 bits only: a stream whose walks never fall into step, which dec_wide_settle
gives up and dec_wide_fn_* decode by transfer functions (round 3: one workgroup, 0.048 GB/s).  For DESIGN.md."""
import ctypes as C
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "tests"))
import harness  # noqa: E402
import parity_cases  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "hpack_lengths"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4 << 20
lengths = [l for count, l in parity_cases.CODER_PROFILES[name] for _ in range(count)]
pats, lens_list = parity_cases.canonical_code(lengths)
patterns, lens = (C.c_uint32 * 256)(*pats), (C.c_uint8 * 256)(*lens_list)
lib = harness.load_product()
coder = lib.aws_huffman_amd_table_coder_new(patterns, lens)
eng = harness.Engine(lib, coder)
data = (32 + harness.splitmix64_bytes(9, n) % 95).astype(np.uint8)
d_in, d_enc, d_back = eng.alloc(n), eng.alloc(4 * n + 64), eng.alloc(n + 64)
eng.upload(d_in, data)
ep = eng.encode_plan([dict(in_offset=0, in_len=n, out_offset=0, out_capacity=4 * n)])
eng.encode_launch(ep, d_in, d_enc)
(rc, err, consumed, produced, _, _), = eng.encode_results(ep, 1)
assert rc == 0 and consumed == n
t0 = time.perf_counter()
eng.encode_launch(ep, d_in, d_enc)
eng.sync()
t_enc = time.perf_counter() - t0
dp = eng.decode_plan([dict(in_offset=0, in_len=produced, first_bit=0, out_offset=0, out_capacity=n)])
# (an untimed launch first: since round 5 the kernels are one code object a path, loaded at the first launch that needs it --
# ~0.6 ms once a process, which a first timed decode launch used to be spared by the encode launch in front of it)
eng.decode_launch(dp, d_enc, d_back)
eng.sync()
t0 = time.perf_counter()
eng.decode_launch(dp, d_enc, d_back)
eng.sync()
t_dec = time.perf_counter() - t0
(rc, err, got, bits), = eng.decode_results(dp, 1)
assert (rc, got) == (0, n), (rc, err, got)
assert np.array_equal(eng.download(d_back, n), data)
print("%s: one stream of %d printable symbols (%d encoded bytes): encode %.2f ms, decode %.1f ms = %.1f MB/s of symbols" % (
    name, n, produced, t_enc * 1e3, t_dec * 1e3, n / t_dec / 1e6))
