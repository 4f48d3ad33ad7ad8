#!/usr/bin/env python3
"""INTEGRATION.md says plan launches compose with HIP graphs: capture an encode launch and a decode launch of its output
into ONE graph on a stream of the caller's, replay the graph over scrambled outputs, and check every replay against the
direct launches.  Prints what a replay costs beside the direct launches (small batches are launch-bound)."""
import ctypes as C
import hashlib
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "tests"))
import harness  # noqa: E402

lib = harness.load_product()
hip = C.CDLL("libamdhip64.so")
patterns, lens = harness.load_table()
coder = lib.aws_huffman_amd_table_coder_new(patterns, lens)
eng = harness.Engine(lib, coder)


def check(rc, what):
    assert rc == 0, "%s failed: hip error %d" % (what, rc)


# (the third case: replays that take the ways BACK every time -- an encoder whose one-pass kernel gives up half-way, so that the
#  ordered kernel behind it does the launch over, and a stream of one repeated symbol, every chunk of which is listed for the
#  kernels behind the regular ones: what those leave in the plans' control words and counters must not outlive a replay)
shortest = int(np.argmin(np.where(np.array(lens) > 0, np.array(lens), 99)))
with harness.encode_road(lib, "one-pass-fails"):
    eng_back = harness.Engine(lib, coder)
for n_items, item_len, ways_back in ((64, 16384, False), (1, 64 << 20, False), (1, 12 << 20, True)):
    n = n_items * item_len
    cap = item_len * 2 + 64
    if ways_back:
        eng = eng_back
    d_in, d_enc, d_back = eng.alloc(n), eng.alloc(n_items * cap), eng.alloc(n + 64)
    ep = eng.encode_plan([dict(in_offset=i * item_len, in_len=item_len, out_offset=i * cap, out_capacity=cap) for i in range(n_items)])
    if ways_back:
        eng.fill(d_in, shortest, n)
    else:
        eng.fill_splitmix64(d_in, n, 11)
    eng.encode_launch(ep, d_in, d_enc)
    res = eng.encode_results(ep, n_items)
    assert all(r[0] == 0 for r in res)
    dp = eng.decode_plan([dict(in_offset=i * cap, in_len=res[i][3], out_offset=i * item_len, out_capacity=item_len) for i in range(n_items)])
    stream = C.c_void_p()
    check(hip.hipStreamCreate(C.byref(stream)), "hipStreamCreate")
    graph, graph_exec = C.c_void_p(), C.c_void_p()
    check(hip.hipStreamBeginCapture(stream, 0), "hipStreamBeginCapture")  # hipStreamCaptureModeGlobal
    rc1 = lib.aws_huffman_amd_encode_plan_launch_staged(ep, d_in, d_enc, False, stream, None)
    rc2 = lib.aws_huffman_amd_decode_plan_launch_staged(dp, d_enc, d_back, stream, None)
    check(hip.hipStreamEndCapture(stream, C.byref(graph)), "hipStreamEndCapture")
    assert rc1 == 0 and rc2 == 0, (rc1, rc2, lib.aws_last_error())
    check(hip.hipGraphInstantiate(C.byref(graph_exec), graph, None, None, 0), "hipGraphInstantiate")
    nodes = C.c_size_t(0)
    check(hip.hipGraphGetNodes(graph, None, C.byref(nodes)), "hipGraphGetNodes")
    for seed in (12, 13, 14):
        # (the same input every time -- the decode plan holds the encoded lengths --, but nothing of the last run left)
        eng.fill_splitmix64(d_enc, n_items * cap, seed)
        eng.fill(d_back, 0xA5, n)
        eng.sync()
        check(hip.hipGraphLaunch(graph_exec, stream), "hipGraphLaunch")
        check(hip.hipStreamSynchronize(stream), "hipStreamSynchronize")
        got_enc = [eng.download(d_enc, res[i][3], offset=i * cap).tobytes() for i in range(min(n_items, 4))]
        got = hashlib.sha256(eng.download(d_back, n).tobytes()).hexdigest()
        want = hashlib.sha256(eng.download(d_in, n).tobytes()).hexdigest()
        assert got == want, "replay %d did not decode back to its input" % seed
        eng.encode_launch(ep, d_in, d_enc)  # the direct launch on the same input
        eng.sync()
        assert got_enc == [eng.download(d_enc, res[i][3], offset=i * cap).tobytes() for i in range(min(n_items, 4))]
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        hip.hipGraphLaunch(graph_exec, stream)
    hip.hipStreamSynchronize(stream)
    t_graph = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.aws_huffman_amd_encode_plan_launch_staged(ep, d_in, d_enc, False, stream, None)
        lib.aws_huffman_amd_decode_plan_launch_staged(dp, d_enc, d_back, stream, None)
    hip.hipStreamSynchronize(stream)
    t_direct = (time.perf_counter() - t0) / reps
    print("%d items of %d bytes%s: encode + decode as one graph of %d nodes, 3 replays over scrambled outputs decoded back bit-exact; "
          "a replay %.1f us, the two direct launches %.1f us" % (n_items, item_len, " (the ways back)" if ways_back else "", nodes.value,
                                                                  t_graph * 1e6, t_direct * 1e6), flush=True)
    hip.hipGraphExecDestroy(graph_exec)
    hip.hipGraphDestroy(graph)
    hip.hipStreamDestroy(stream)
