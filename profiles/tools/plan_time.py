#!/usr/bin/env python3
"""What MAKING a plan costs (every quoted rate assumes one that exists): aws_huffman_amd_{en,de}code_plan_new + _destroy
(device allocations included) and _reset (the plan's arrays reused: the fill alone) for BASELINE configs[3]'s 65 536
buffers and for 1 Mi header-sized items, best of five, on the box's host + GPU; and aws_huffman_amd_decode_plan_from_encode
for the latter (the decode plan made on the device from the encode launch's records).  Prints JSON."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np

import harness

lib = harness.load_product(sys.argv[1] if len(sys.argv) > 1 else None)
coder = lib.aws_huffman_amd_table_coder_new(*harness.load_table())
eng = harness.Engine(lib, coder, device=0)
rng = np.random.default_rng(1)


def best(fn, n=5):
    out = 1e9
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        out = min(out, time.perf_counter() - t0)
    return round(out * 1e3, 3)


def decode_items(lens, cap):
    arr = (harness.AmdDecodeItem * len(lens))()
    off = oo = 0
    for i, l in enumerate(lens):
        arr[i].in_offset, arr[i].in_len, arr[i].out_offset, arr[i].out_capacity, arr[i].first_bit = off, int(l), oo, cap, 0
        off += int(l)
        oo += cap
    return arr


def encode_items(lens, cap):
    arr = (harness.AmdEncodeItem * len(lens))()
    off = oo = 0
    for i, l in enumerate(lens):
        arr[i].in_offset, arr[i].in_len, arr[i].out_offset, arr[i].out_capacity = off, int(l), oo, cap
        off += int(l)
        oo += cap
    return arr


def measure(kind, arr, n):
    new = getattr(lib, "aws_huffman_amd_%s_plan_new" % kind)
    reset = getattr(lib, "aws_huffman_amd_%s_plan_reset" % kind)
    destroy = getattr(lib, "aws_huffman_amd_%s_plan_destroy" % kind)

    def make():
        plan = C.c_void_p()
        assert new(C.byref(plan), eng.h, arr, n) == 0
        destroy(plan)

    plan = C.c_void_p()
    assert new(C.byref(plan), eng.h, arr, n) == 0
    out = {"new_and_destroy_ms": best(make), "reset_ms": best(lambda: reset(plan, arr, n))}
    destroy(plan)
    return out


out = {
    "decode_cfg4_65536_items_of_19_KB": measure("decode", decode_items(rng.integers(19300, 19500, 65536), 16384), 65536),
    "encode_cfg4_65536_items_of_16_KiB": measure("encode", encode_items([16384] * 65536, 20480), 65536),
    "decode_1Mi_items_of_16_to_80_B": measure("decode", decode_items(rng.integers(16, 81, 1 << 20), 128), 1 << 20),
    "encode_1Mi_items_of_16_to_80_B": measure("encode", encode_items(rng.integers(16, 81, 1 << 20), 128), 1 << 20),
}


def chained(lens, cap):
    """the decode plan of what an encode launch of these items left, made on the device from its records (the call alone: it
    queues one small kernel); and encode launch -> chained plan -> decode launch -> both done, without the lengths on the host"""
    n = len(lens)
    arr = encode_items(lens, cap)
    eplan, dplan = C.c_void_p(), C.c_void_p()
    assert lib.aws_huffman_amd_encode_plan_new(C.byref(eplan), eng.h, arr, n) == 0
    assert lib.aws_huffman_amd_decode_plan_new(C.byref(dplan), eng.h, None, 0) == 0
    total_in, total_out = int(sum(int(l) for l in lens)) + 64, n * cap + 64
    d_in, d_enc, d_back = eng.alloc(total_in), eng.alloc(total_out), eng.alloc(total_in)
    eng.fill_splitmix64(d_in, total_in, 3)
    lib.aws_huffman_amd_decode_plan_from_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]

    def chain():
        assert lib.aws_huffman_amd_decode_plan_from_encode(dplan, eplan, None) == 0

    def whole():
        eng.encode_launch(eplan, d_in, d_enc)
        chain()
        eng.decode_launch(dplan, d_enc, d_back)
        eng.sync()

    eng.encode_launch(eplan, d_in, d_enc)
    eng.sync()
    res = {"from_encode_call_ms": best(chain), "encode_chain_decode_sync_ms": best(whole)}
    assert np.array_equal(eng.download(d_back, total_in - 64), eng.download(d_in, total_in - 64))
    lib.aws_huffman_amd_encode_plan_destroy(eplan)
    lib.aws_huffman_amd_decode_plan_destroy(dplan)
    for ptr in (d_in, d_enc, d_back):
        eng.free(ptr)
    return res


out["decode_1Mi_items_chained_to_their_encode_plan"] = chained(rng.integers(16, 81, 1 << 20), 128)
print(json.dumps({"plan_ms": out}))
