"""TEST INFRASTRUCTURE.  Parity scenarios against the emulator build of the library made with AddressSanitizer on top
of UBSan (tests/test_emulated_asan.py builds it and runs this with libasan preloaded): the roads that index by what a
stream says -- one-launch calls, long-code items across blocks, fixed-length coders, the ways back -- with every
access of the kernels' code checked."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import harness  # noqa: E402
import parity_cases as pc  # noqa: E402

w = pc.World(harness.oracle_codec(), harness.Codec(harness.load_product(sys.argv[1]), "aws_"))
import time

for run in (lambda: pc.block_decode_calls(w, wants=(130, 513, 8192, 8193, 16385, 32768), kinds=("uniform", "long")), lambda: pc.fixed_length_coders(w), lambda: pc.wide_long_code_items(w, modes=(None,)),
            lambda: pc.one_shot_roundtrips(w, sizes=[255, 4097, 16384])):
    t0 = time.time()
    run()
    print("%.0f s" % (time.time() - t0), flush=True)
print("no finding")
