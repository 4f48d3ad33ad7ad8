#!/usr/bin/env python3
"""The parity scenarios of tests/parity_cases.py again and again with fresh seeds, on the GPU, for as long as asked:
every result is compared with the CPU oracle, bit for bit.  Not part of the timed or graded runs: a soak for the
paths the seeded tests visit once.

  usage: fuzz_rounds.py [seconds=240] [first_seed=1000] [only-rounds-named-like]
"""
import os
import sys
import time
import traceback

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "tests"))
import harness  # noqa: E402
import parity_cases as pc  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
only = sys.argv[3] if len(sys.argv) > 3 else ""  # a substring: only the rounds whose names hold it
lib = harness.load_product()
assert lib.aws_huffman_amd_device_count() >= 1
w = pc.World(harness.oracle_codec(), harness.Codec(lib, "aws_"))
eng = harness.Engine(lib, w.pcoder)
rounds = [
    ("tiny_encode_items", lambda s: pc.tiny_encode_items(w, n_items=600, seed=s, engine=eng)),
    ("tiny_encode_items holes", lambda s: pc.tiny_encode_items(w, n_items=600, seed=s, holes=True)),
    ("tiny_decode_items", lambda s: pc.tiny_decode_items(w, n_items=600, seed=s, engine=eng)),
    ("tiny_decode_items hpack", lambda s: pc.tiny_decode_items(w, n_items=600, seed=s, profile="hpack_lengths")),
    ("tiny_encode_items many", lambda s: pc.tiny_encode_items(w, n_items=11000, seed=s, engine=eng) if s % 4 == 0 else None),
    ("tiny_decode_items many", lambda s: pc.tiny_decode_items(w, n_items=4000, seed=s, engine=eng) if s % 4 == 1 else None),
    ("batched_device_api", lambda s: pc.batched_device_api(w, n_items=24, seed=s, engine=eng)),
    ("batched_device_api 3000", lambda s: pc.batched_device_api(w, n_items=14, seed=s, item_len=3000, engine=eng)),
    ("garbage_decode", lambda s: pc.garbage_decode(w, seed=s, rounds=40)),
    ("streaming_encode", lambda s: pc.streaming_encode(w, [700, 40000], seed=s)),
    ("streaming_decode", lambda s: pc.streaming_decode(w, [700, 40000], seed=s)),
    ("one_shot_roundtrips", lambda s: pc.one_shot_roundtrips(w, [1, 513, 16385, 100001], seed=s)),
    ("cut_streams", lambda s: pc.cut_streams(w, seed=s, chunks=(1, 3), step=13, span=70, n=120_000)),
    ("other_coders", lambda s: pc.other_coders(w, n=30000, seed=s)),
    ("block_decode_calls", lambda s: pc.block_decode_calls(w, seed=s) if s % 2 == 0 else None),
    ("wide_long_code_items", lambda s: pc.wide_long_code_items(w, seed=s) if s % 3 == 0 else None),
    ("fixed_length_coders", lambda s: pc.fixed_length_coders(w, seed=s) if s % 3 == 1 else None),
    ("damaged_long_streams", lambda s: pc.damaged_long_streams(w, seed=s) if s % 8 == 0 else None),  # 10 M symbols: now and then
    # round 4: several short end-of-stream chunks a workgroup (dec_sync_pack, dec_emit_pack), plans made on the device
    ("mid_sized_items", lambda s: pc.mid_sized_items(w, n_items=120, seed=s, engine=eng, modes=(None,))),
    ("mid_sized_items narrow", lambda s: pc.mid_sized_items(w, n_items=90, seed=s, engine=eng, modes=(None,), longest=1900)),
    ("many_header_sized_items", lambda s: pc.many_header_sized_items(w, n_items=4200, seed=s, engine=eng) if s % 2 == 0 else None),
    ("null_empty_cursors", lambda s: pc.null_empty_cursors(w, seed=s) if s % 4 == 0 else None),
    # later in round 4: streams whose walks never become one (dec_sync_few / _true, dec_wide_fn_*), plans chained on the device
    ("streams_out_of_step", lambda s: pc.streams_out_of_step(w, n=260_000 + 996 * (s % 50), seed=s, modes=(None,))),
    ("never_in_step_stream", lambda s: pc.never_in_step_stream(w, n=300_000 + 1009 * (s % 40), seed=s) if s % 3 == 2 else None),
    ("encode_then_decode_on_the_device", lambda s: pc.encode_then_decode_on_the_device(w, seed=s, engine=eng, batches=((60, 60), (7300, 90)))
     if s % 2 == 1 else None),
    ("plans_one_after_another", lambda s: pc.plans_one_after_another(w, seed=s) if s % 4 == 2 else None),
    # round 5: dec_sync_one's rare lanes, plans made on the device, items a wave encodes without segments (one tile; up
    # to a segment in a plan of 256 items or more), streams with two last chunks
    ("walks_that_never_meet", lambda s: pc.walks_that_never_meet(w, seed=s, engine=eng, runs=(130 + s % 97, 260 + s % 211, 700 + s % 409))),
    ("mid_sized_encode_items", lambda s: pc.tiny_encode_items(w, n_items=300 + s % 40, seed=s, engine=eng, max_len=20000, edge_lens=False,
                                                              more_lens=(4095, 4096, 4097, 8192, 12288, 16383, 16384, 16385))),
    ("mid_sized_encode_items few", lambda s: pc.tiny_encode_items(w, n_items=120, seed=s, engine=eng, max_len=9000, edge_lens=False,
                                                                  more_lens=(4095, 4096, 4097, 8192))),
    ("plans_made_on_the_device", lambda s: pc.plans_made_on_the_device(w, seed=s, engine=eng, big=600_000 + 1013 * (s % 60), n_small=200 + s % 120)
     if s % 3 == 0 else None),
    ("streams_with_two_last_chunks", lambda s: pc.streams_with_two_last_chunks(w, seed=s, engine=eng, modes=(None,)) if s % 3 == 1 else None),
    # round 6: a few stream ends folded into the sync kernel's grid (the stream's last symbols followed by the workgroup), quiet
    # plans (no kernels for listed chunks until a fetch says chunks were listed), the encoder's way back by ticket
    ("few_ends_among_many_chunks", lambda s: pc.few_ends_among_many_chunks(w, n=330_000 + 1017 * (s % 70), seed=s, engine=eng, modes=(None,))
     if s % 2 == 0 else None),
    ("quiet_plans", lambda s: pc.quiet_plans(w, n=300_000 + 997 * (s % 50), seed=s, engine=eng) if s % 3 == 2 else None),
    ("encode_roads", lambda s: pc.encode_roads(w, sizes=(200_000 + 1021 * (s % 30), 16384, 40_000 + s % 999, 1_000_000 + 4099 * (s % 20)), seed=s)
     if s % 4 == 3 else None),
]
rounds = [r for r in rounds if only in r[0]]
t0 = time.time()
done = failed = 0
while time.time() - t0 < budget:
    for name, run in rounds:
        try:
            run(seed)
        except Exception:  # noqa: BLE001
            failed += 1
            print("FAILED %s seed %d" % (name, seed))
            traceback.print_exc()
        done += 1
        if time.time() - t0 >= budget:
            break
    seed += 1
print("%d scenario runs, %d failed, seeds up to %d, %.0f s" % (done, failed, seed, time.time() - t0))
sys.exit(1 if failed else 0)
