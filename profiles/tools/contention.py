#!/usr/bin/env python3
"""Two engines on ONE device, each on a host thread of its own, launching one-pass encodes at the same time: the static
tile schedule of enc_onepass wants the whole grid resident, two grids at once are not, so look-back waits run out -- and
what every launch leaves on its stream must still be the right bytes (the three-kernel road, queued behind a device-side
gate in the same launch, does it over).  Every launch's output is read back BEFORE its results and compared with an
undisturbed run's digest; prints how many launches of each engine gave up."""
import hashlib
import os
import sys
import threading

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "tests"))
import harness  # noqa: E402

lib = harness.load_product()
patterns, lens = harness.load_table()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256 << 20
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 12


class Worker:
    def __init__(self, seed):
        # (a coder object each: engines are cached per coder and device)
        self.coder = lib.aws_huffman_amd_table_coder_new(patterns, lens)
        self.eng = harness.Engine(lib, self.coder)
        self.d_in, self.d_enc = self.eng.alloc(n), self.eng.alloc(2 * n + 64)
        self.eng.fill_splitmix64(self.d_in, n, seed)
        self.plan = self.eng.encode_plan([dict(in_offset=0, in_len=n, out_offset=0, out_capacity=2 * n)])
        self.eng.encode_launch(self.plan, self.d_in, self.d_enc)
        (rc, _, consumed, self.e_len, _, _), = self.eng.encode_results(self.plan, 1)
        assert rc == 0 and consumed == n
        self.want = self.digest()
        self.gave_up = self.wrong = 0
        # and the same with decode launches (no kernel of theirs waits for another workgroup: nothing can run out)
        self.d_back = self.eng.alloc(n + 64)
        self.dplan = self.eng.decode_plan([dict(in_offset=0, in_len=self.e_len, out_offset=0, out_capacity=n)])
        self.want_back = self.digest_of(self.d_in, n)
        self.dec_gave_up = self.dec_wrong = 0

    def digest(self):
        h = hashlib.sha256()
        for off in range(0, self.e_len, 64 << 20):
            h.update(self.eng.download(self.d_enc, min(64 << 20, self.e_len - off), offset=off).tobytes())
        return h.hexdigest()

    def digest_of(self, ptr, size):
        h = hashlib.sha256()
        for off in range(0, size, 64 << 20):
            h.update(self.eng.download(ptr, min(64 << 20, size - off), offset=off).tobytes())
        return h.hexdigest()

    def run_decode(self, barrier):
        for _ in range(rounds):
            self.eng.fill(self.d_back, 0x5A, n)
            self.eng.sync()
            barrier.wait()
            self.eng.decode_launch(self.dplan, self.d_enc, self.d_back)
            got = self.digest_of(self.d_back, n)
            (rc, _, symbols, _), = self.eng.decode_results(self.dplan, 1)
            self.dec_wrong += (rc, symbols, got) != (0, n, self.want_back)
            self.dec_gave_up += self.eng.decode_road(self.dplan) == 2

    def run(self, barrier):
        for _ in range(rounds):
            self.eng.fill(self.d_enc, 0x5A, self.e_len)
            self.eng.sync()
            barrier.wait()
            self.eng.encode_launch(self.plan, self.d_in, self.d_enc)
            got = self.digest()  # the stream as the launch leaves it: read behind it, before the records
            (rc, _, consumed, e_len, _, _), = self.eng.encode_results(self.plan, 1)
            self.wrong += (rc, consumed, e_len, got) != (0, n, self.e_len, self.want)
            self.gave_up += self.eng.encode_road(self.plan) == 2


workers = [Worker(21), Worker(22)]
barrier = threading.Barrier(len(workers))
threads = [threading.Thread(target=w.run, args=(barrier,)) for w in workers]
for t in threads:
    t.start()
for t in threads:
    t.join()
print("two engines on one device, %d simultaneous one-pass encodes of %d MiB each: wrong outputs %s, launches that gave up and "
      "were done over on the device %s" % (rounds, n >> 20, [w.wrong for w in workers], [w.gave_up for w in workers]))
threads = [threading.Thread(target=w.run_decode, args=(barrier,)) for w in workers]
for t in threads:
    t.start()
for t in threads:
    t.join()
print("the same with simultaneous decode launches: wrong outputs %s" % ([w.dec_wrong for w in workers],))
sys.exit(1 if any(w.wrong or w.dec_wrong for w in workers) else 0)
