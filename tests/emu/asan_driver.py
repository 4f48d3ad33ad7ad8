"""TEST INFRASTRUCTURE.  Parity scenarios against the emulator build of the library made with AddressSanitizer on top
of UBSan (tests/test_emulated_asan.py builds it and runs this with libasan preloaded): the roads that index by what a
stream says -- one-launch calls, long-code items across blocks (settled and by transfer functions), fixed-length coders,
the ways back, the chunks whose walks never become one, the packed end-of-stream chunks, plans chained on the device, the
encoder's wave-per-item road, a long stream's end inside the sync kernel's grid, the encoder's way back by ticket --
with every access of the kernels' code checked."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import harness  # noqa: E402
import parity_cases as pc  # noqa: E402

w = pc.World(harness.oracle_codec(), harness.Codec(harness.load_product(sys.argv[1]), "aws_"))
import time

for run in (lambda: pc.block_decode_calls(w, wants=(130, 513, 8192, 8193, 16385, 32768), kinds=("uniform", "long")), lambda: pc.fixed_length_coders(w),
            lambda: pc.wide_long_code_items(w, modes=(None, "wide-fails")),  # (round 4: and every stream by transfer functions, dec_wide_fn_*)
            lambda: pc.one_shot_roundtrips(w, sizes=[255, 4097, 16384]),
            # round 4: streams whose walks never become one (dec_sync_few / _true), several short chunks a workgroup, plans chained on the device
            lambda: pc.streams_out_of_step(w, n=170_000, modes=(None,)), lambda: pc.never_in_step_stream(w, n=200_000),
            lambda: pc.mid_sized_items(w, n_items=40, modes=(None,)), lambda: pc.encode_then_decode_on_the_device(w, batches=((40, 50),)),
            # round 5: dec_sync_one's rare lanes (walks that never meet: the lane behind walks again from memory), encode items a
            # wave takes without segments (one tile; up to a segment in a plan of 256 items: the capacity edge found by the packing
            # wave, staged through the image's LDS)
            lambda: pc.walks_that_never_meet(w, runs=(130, 420)),
            lambda: pc.tiny_encode_items(w, n_items=270, seed=161, max_len=20000, edge_lens=False, more_lens=(4095, 4096, 4097, 16383, 16384, 16385)),
            lambda: pc.tiny_encode_items(w, n_items=90, seed=162, max_len=9000, edge_lens=False, more_lens=(4095, 4096, 4097)),
            # round 6: a few stream ends as workgroups of the sync kernel's own grid (the stream's last symbols followed there),
            # the encoder's way back by ticket, dec_sync_one's rows through shifted words
            lambda: pc.few_ends_among_many_chunks(w, modes=(None,)),
            lambda: pc.encode_roads(w, sizes=(200_000, 16384, 40_000))):
    t0 = time.time()
    run()
    print("%.0f s" % (time.time() - t0), flush=True)
print("no finding")
