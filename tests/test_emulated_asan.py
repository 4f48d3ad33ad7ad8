"""The product's kernels compiled for the CPU with AddressSanitizer + UBSan (GPU sanitizers are not available on the
pool): the parity scenarios of the roads added in rounds 3 and 4, every load and store of the kernels' code checked."""
import os
import subprocess
import sys
import tempfile

import harness

EMU_DIR = os.path.join(harness.REPO, "tests", "emu")
# (outside the repository: 60 MB of objects that a GPU box, which gets a snapshot of the tree, has no use for)
BUILD = os.path.join(tempfile.gettempdir(), "aws-c-compression-emu-asan-%d" % os.getuid())
ASAN_SO = os.path.join(BUILD, "libaws-c-compression-emu-asan.so")


def test_roads_under_address_sanitizer():
    subprocess.check_call(
        ["make", "-s", "-C", EMU_DIR, "BUILD=" + BUILD, "TARGET=" + ASAN_SO,
         "SAN=-fsanitize=address,undefined -fno-sanitize-recover=undefined"], stdout=subprocess.DEVNULL)
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    # (the emulator's work-items are ucontext fibers: the stack-use-after-return mode does not know them)
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0")
    done = subprocess.run([sys.executable, os.path.join(EMU_DIR, "asan_driver.py"), ASAN_SO], env=env, capture_output=True,
                          text=True, timeout=2400)
    assert done.returncode == 0, (done.stdout + done.stderr)[-4000:]
    assert "no finding" in done.stdout
