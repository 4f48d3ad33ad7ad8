#!/usr/bin/env python3
"""Batches of HPACK-header-sized items (the reference's real workload: tens of bytes per call), device resident,
one plan: 1 Mi items of 16..80 bytes.  For DESIGN.md.

  usage: tiny_items.py [profile]     profile: one of tests/parity_cases.py CODER_PROFILES (e.g. hpack_lengths: codes of
                                     5..30 bits) instead of the reference's test coder"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "tests"))
import harness  # noqa: E402

lib = harness.load_product()
patterns, lens = harness.load_table()
if len(sys.argv) > 1:
    import ctypes as C

    import parity_cases

    lengths = [l for count, l in parity_cases.CODER_PROFILES[sys.argv[1]] for _ in range(count)]
    pats, lens_list = parity_cases.canonical_code(lengths)
    patterns, lens = (C.c_uint32 * 256)(*pats), (C.c_uint8 * 256)(*lens_list)
coder = lib.aws_huffman_amd_table_coder_new(patterns, lens)
oracle = harness.oracle_codec()
ocoder = oracle.lib.oracle_table_coder_new(patterns, lens)
eng = harness.Engine(lib, coder)
rng = np.random.default_rng(3)
items = 1 << 20
sizes = rng.integers(16, 81, items)
offs = np.concatenate(([0], np.cumsum(sizes)))
total = int(offs[-1])
raw = harness.splitmix64_bytes(4, total)
data = (32 + raw % 95).astype(np.uint8)
d_in = eng.alloc(total)
eng.upload(d_in, data)
cap = 320  # bytes of output room per item (>= 80 * 30 / 8)
d_enc = eng.alloc(items * cap)
t0 = time.perf_counter()
ep = eng.encode_plan([dict(in_offset=int(offs[i]), in_len=int(sizes[i]), out_offset=i * cap, out_capacity=cap) for i in range(items)])
t_plan = time.perf_counter() - t0
eng.encode_launch(ep, d_in, d_enc)
res = eng.encode_results(ep, items)
assert all(r[0] == 0 for r in res)
t0 = time.perf_counter()
for _ in range(3):
    eng.encode_launch(ep, d_in, d_enc)
eng.sync()
t_enc = (time.perf_counter() - t0) / 3
d_back = eng.alloc(total + 64)
dp = eng.decode_plan([dict(in_offset=i * cap, in_len=res[i][3], out_offset=int(offs[i]), out_capacity=int(sizes[i])) for i in range(items)])
eng.decode_launch(dp, d_enc, d_back)
dres = eng.decode_results(dp, items)
assert all(r[0] == 0 and r[2] == int(sizes[i]) for i, r in enumerate(dres)), [(i, r) for i, r in enumerate(dres) if r[0] != 0][:3]
assert np.array_equal(eng.download(d_back, total), data)
# spot check against the oracle
for i in (0, 1, 777, items - 1):
    want = oracle.encode_all(ocoder, data[offs[i]:offs[i + 1]])
    got = eng.download(d_enc, res[i][3], offset=i * cap)
    assert np.array_equal(got, want), i
t0 = time.perf_counter()
for _ in range(3):
    eng.decode_launch(dp, d_enc, d_back)
eng.sync()
t_dec = (time.perf_counter() - t0) / 3
print((sys.argv[1] if len(sys.argv) > 1 else "test coder") + ": %d items of 16..80 printable bytes (%.1f MiB): encode %.2f ms = %.1f M items/s = %.1f GiB/s; decode %.2f ms = %.1f M items/s = %.1f GiB/s of symbols (plan built on the host in %.1f s)" % (
    items, total / 2**20, t_enc * 1e3, items / t_enc / 1e6, total / 2**30 / t_enc, t_dec * 1e3, items / t_dec / 1e6, total / 2**30 / t_dec, t_plan))
