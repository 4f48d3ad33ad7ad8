"""The reference's three fuzz properties (reference tests/fuzz/decode.c, transitive.c, transitive_chunked.c) as plain C
programs over a seeded corpus (tests/fuzz/), built with AddressSanitizer + UBSan against the emulator build of the
library -- the product's kernels compiled for the CPU with UBSan (GPU sanitizers are not available on the pool).
A finding aborts the program; the test is that none does."""
import os
import subprocess

import pytest

import harness

FUZZ_DIR = os.path.join(harness.REPO, "tests", "fuzz")


@pytest.fixture(scope="module")
def built():
    subprocess.check_call(["make", "-s", "-C", os.path.join(harness.REPO, "tests", "emu")], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-s", "-C", FUZZ_DIR], stdout=subprocess.DEVNULL)
    return os.path.join(FUZZ_DIR, "build")


@pytest.mark.parametrize("target,inputs,max_len,seed", [
    ("fuzz_decode", 48, 1500, 11),              # F1: arbitrary bytes into a fresh decoder
    ("fuzz_transitive", 40, 1500, 12),          # F2: every byte string round-trips
    ("fuzz_transitive_chunked", 16, 500, 13),   # F3: the same, output offered 1 .. 128 bytes at a time
])
def test_seeded_corpus_has_no_finding(built, target, inputs, max_len, seed):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")  # (the coder and the engines live as long as the process)
    done = subprocess.run([os.path.join(built, target), str(inputs), str(max_len), str(seed)], env=env,
                          capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, done.stderr[-3000:]
    assert "no finding" in done.stdout
