"""The N > 1 path of bench.py on CPU: two ranks under torch.distributed.run with the gloo
backend, exactly as the driver launches it on an 8-GPU node, but with the kernels compiled
for the host by the emulator build (tests/emu) since there is no GPU here.

What is checked is the rank logic, not speed: every rank encodes and decodes its OWN stream
(seed 5 + rank, SURVEY.md 8e: independent streams, no collective on the data path), the
per-rank results are bit-exact against the oracle, rank 0 prints exactly one JSON line, and
`value` is the whole-job figure (all ranks' bytes over the max-over-ranks time).
"""
import hashlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import harness
import parity_cases as pc
from test_emulated_kernels import EMU_DIR, EMU_SO

STREAM_BYTES = 160 * 1024 + 13  # ten-and-a-bit encode segments, ragged tail


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def emulator():
    subprocess.check_call(["make", "-s", "-C", EMU_DIR])
    return EMU_SO


def run_bench(emulator, nproc, extra=(), launcher=True):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    for name in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(name, None)
    bench = os.path.join(harness.REPO, "bench.py")
    args = ["--gpus", str(nproc), "--steps", "2", "--warmup", "1", "--bytes", str(STREAM_BYTES),
            "--no-cpu-baseline", "--library", emulator, *extra]
    if nproc == 1 or not launcher:
        cmd = [sys.executable, bench, *args]  # (N > 1 without a launcher: bench.py starts its ranks itself)
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), bench, *args]
    done = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert done.returncode == 0, done.stderr[-4000:]
    lines = [ln for ln in done.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line, got: %r" % done.stdout
    return json.loads(lines[0])


def expected_stream(oracle, rank):
    w_coder = oracle.lib.oracle_table_coder_new(*harness.load_table())
    data = harness.splitmix64_bytes(5 + rank, STREAM_BYTES)
    dst = np.zeros(STREAM_BYTES * 2 + 64, dtype=np.uint8)
    r = oracle.encode_call(oracle.new_encoder(w_coder), data, 0, dst, 0, dst.size)
    assert r.rc == 0
    return r.produced, hashlib.sha256(dst[:r.produced].tobytes()).hexdigest()


def check_line(out, world):
    assert out["n_gpus"] == world and out["steps"] == 2 and out["warmup"] == 1
    assert out["scaling"] == "weak" and out["higher_is_better"] is True and out["vs_baseline"] is None
    assert out["unit"] == "GiB/s" and out["dtype"] == "u8" and out["data"] == "synthetic"
    assert "workload" in out["config"] and "model" not in out["config"]
    # whole-job value: every rank's bytes over the slowest rank's time per step
    per_step_s = out["ms_per_step"] * 1e-3
    assert out["value"] == pytest.approx(world * STREAM_BYTES / 2**30 / per_step_s, rel=2e-2)
    for name in ("roofline", "roofline_encode", "roofline_decode", "roofline_kernel"):
        for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert key in out[name], (name, key)
        assert out[name]["bound"] == "hbm" and out[name]["peak"] == 8000.0
    # the headline roofline is a PATH: all its kernels, N + E bytes
    assert out["roofline"]["path"] in ("encode", "decode") and len(out["roofline"]["kernels"]) == 3
    assert out["roofline"]["algorithmic_bytes_per_launch"] == STREAM_BYTES + out["ranks"][0]["encoded_bytes"]


def test_two_ranks_independent_streams(emulator, oracle):
    out = run_bench(emulator, 2)
    check_line(out, 2)
    assert [r["rank"] for r in out["ranks"]] == [0, 1]
    assert [r["seed"] for r in out["ranks"]] == [5, 6]
    for r in out["ranks"]:
        e_len, digest = expected_stream(oracle, r["rank"])
        assert (r["encoded_bytes"], r["sha256_encoded"]) == (e_len, digest), "rank %d stream differs from the oracle" % r["rank"]
    assert out["ranks"][0]["sha256_encoded"] != out["ranks"][1]["sha256_encoded"]
    assert out["cpu_baseline"] is None  # reported at N = 1 only


def test_two_ranks_without_a_launcher(emulator, oracle):
    """`python bench.py --gpus 2` as a plain process, the way the driver runs `--gpus 1`: two ranks all the same."""
    out = run_bench(emulator, 2, launcher=False)
    check_line(out, 2)
    assert [r["rank"] for r in out["ranks"]] == [0, 1] and [r["seed"] for r in out["ranks"]] == [5, 6]
    assert [r["device"] for r in out["ranks"]] == [0, 0]  # (the emulator has one device; on a node: 0 and 1)
    for r in out["ranks"]:
        e_len, digest = expected_stream(oracle, r["rank"])
        assert (r["encoded_bytes"], r["sha256_encoded"]) == (e_len, digest), "rank %d stream differs from the oracle" % r["rank"]


def expected_batch(oracle, rank, world, buffers, size):
    """sha256 over the complete encoded streams of the buffers rank `rank` takes (i = rank mod world), in order."""
    coder = oracle.lib.oracle_table_coder_new(*harness.load_table())
    h, total = hashlib.sha256(), 0
    for i in range(rank, buffers, world):
        enc = oracle.encode_all(coder, harness.splitmix64_bytes(2 + i, size))
        h.update(enc.tobytes())
        total += enc.size
    return total, h.hexdigest()


def test_two_ranks_split_the_batch(emulator, oracle):
    """BASELINE configs[3] over two ranks: buffer i on rank i mod 2, every fourth buffer of the batch
    capacity-limited and resumed; each rank's encoded streams against the oracle's."""
    buffers, size = 22, 16384
    out = run_bench(emulator, 2, extra=("--workload", "cfg4", "--buffers", str(buffers), "--buffer-bytes", str(size)))
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["workload"] == "cfg4"
    assert out["value"] == pytest.approx(buffers * size / 2**30 / (out["ms_per_step"] * 1e-3), rel=2e-2)
    assert [r["buffers"] for r in out["ranks"]] == [11, 11]
    for r in out["ranks"]:
        e_len, digest = expected_batch(oracle, r["rank"], 2, buffers, size)
        assert (r["encoded_bytes"], r["sha256_encoded_streams"]) == (e_len, digest), "rank %d differs from the oracle" % r["rank"]


def test_single_rank_line(emulator, oracle):
    """The default line of one GPU: the stream, verified after the timed steps, with BASELINE configs[3] and the
    reference's entry points on host memory riding along as extra keys (here with a small batch)."""
    out = run_bench(emulator, 1, extra=("--buffers", "12", "--buffer-bytes", "16384", "--header-items", "3000"))
    check_line(out, 1)
    e_len, digest = expected_stream(oracle, 0)
    assert (out["config"]["encoded_bytes"], out["config"]["sha256_encoded"]) == (e_len, digest)
    assert out["config"]["bit_exact"] is True and "after the last timed step" in out["config"]["bit_exact_checked"]
    assert out["config"]["encode_road"] in ("one-pass", "three-kernel") and out["config"]["decode_road"] == "two-pass"
    for leg in ("cfg4", "mid_items", "host_abi"):
        assert out[leg]["bit_exact"] is True and "value_GiBps" in out[leg] and "encode_ms" in out[leg], leg  # (the emulator has no clock for events)
    assert "configs[3]" in out["cfg4"]["workload"] and "HOST memory" in out["host_abi"]["workload"]
    # a batch of header-sized items encoded, its decode plan chained to it on the device, decoded back
    assert out["header_items"]["bit_exact"] is True and out["header_items"]["items_per_s"] > 0 and "3000 items" in out["header_items"]["workload"]
    # what making the batch's plans cost rides along, and a traffic figure only where it was measured for the workload
    assert set(out["cfg4"]["plan_ms"]) == {"encode", "encode_resume", "decode"} and out["cfg4"]["roofline"]["traffic"] is None
