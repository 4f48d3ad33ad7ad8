import os, sys
sys.path.insert(0, "tests")
import numpy as np, harness, parity_cases as pc
lib = harness.load_product()
w = pc.World(harness.oracle_codec(), harness.Codec(lib, "aws_"))
eng = harness.Engine(lib, w.pcoder)
for rep in range(6):
    for s in (810096, 810327, 810579):
        try:
            pc.plans_made_on_the_device(w, seed=s, engine=eng, big=600_000 + 1013 * (s % 60), n_small=200 + s % 120)
            print(s, "ok", flush=True)
        except AssertionError as e:
            print(s, "FAILED", str(e)[:1500], flush=True)
