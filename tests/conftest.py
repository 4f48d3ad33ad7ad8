import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: minutes of CPU; opt in with HUFFMAN_SLOW=1")


@pytest.fixture(scope="session")
def oracle():
    import subprocess

    import harness

    if not os.path.exists(harness.ORACLE_SO):
        subprocess.check_call(["make", "-C", os.path.join(harness.REPO, "oracle"), "libhuffman_oracle.so"])
    return harness.oracle_codec()


@pytest.fixture(scope="session")
def oracle_coder(oracle):
    import harness

    patterns, lens = harness.load_table()
    coder = oracle.lib.oracle_table_coder_new(patterns, lens)
    assert coder
    return coder
