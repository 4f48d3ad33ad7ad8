#!/usr/bin/env python3
"""Benchmark of the Huffman hot path on MI355X.

Contract (one JSON line on stdout from rank 0):

  python bench.py --gpus N --steps K --warmup W [--workload stream|cfg4|host-abi]

  * --workload stream (default) = BASELINE.json configs[1]+[2], the configuration the metric is quoted on: one
    1 GiB stream of splitmix64 bytes per GPU (seed 5 + rank), encoded with the test coder of the reference
    (tests/test_huffman_static_table.def) and decoded back.  One STEP = one encode of the stream + one decode
    of the result, inputs and outputs resident in HBM.  Weak scaling: every GPU has its own stream.
  * --workload cfg4 = configs[3]: 65 536 buffers of 16 KiB (buffer i = splitmix64 seed 2 + i), every fourth one
    capacity-limited (SHORT_BUFFER with the reference's record, then a resume call), decoded back; with N ranks
    buffer i goes to rank i mod N (strong scaling: the batch is fixed).
  * --workload host-abi = the eight reference entry points on HOST pointers (aws_huffman_encode / _decode of one
    256 MiB buffer): PCIe-inclusive, never the headline value.
  * value = GiB of input symbols pushed through encode+decode per second, whole job:
    sum over ranks of the symbols / (t_encode + t_decode), time = max over ranks between barriers.
  * roofline = the slower of the two PATHS (encode, decode), its algorithmic bytes N + E (every input byte read
    once, every output byte written once: SURVEY.md 8d) over the sum of its kernels' durations -- medians over the
    timed steps of HIP events recorded on the stream between the kernels -- against 8 TB/s of HBM.
    roofline_encode / roofline_decode give both paths, roofline_kernel the single longest kernel.
  * cpu_baseline = the CPU oracle (a port of the reference's scalar loop; the reference itself needs aws-c-common
    and cannot be built here) on the host: the same stream on ONE core (the reference is single-threaded per
    stream) on a bounded prefix, and the configs[3] batch on ALL cores (one thread per core), CPU model stated.

For N > 1 the driver launches one process per GPU through torch.distributed.run; ranks share nothing but a barrier
and a max-reduction of the step time (no collective on the data path: the items are independent, DESIGN.md
"Multi-GPU").  Started as a plain process (`python bench.py --gpus N`, WORLD_SIZE unset) it starts the N ranks
itself (spawn_ranks) and passes rank 0's line through.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import statistics
import sys
import threading
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "tests"))

GIB = 1 << 30
HBM_PEAK_BYTES_PER_S = 8.0e12  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
PCIE_PEAK_BYTES_PER_S = 63.0e9  # PCIe Gen5 x16 (spec), same guide
METRIC = "GiB/s input consumed, encode+decode, 1 GiB random bytes; % HBM roofline"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=("stream", "cfg4", "host-abi"), default="stream")
    ap.add_argument("--no-fresh", action="store_true", help="cfg4: without the fresh-batch measurements (profiling runs: only the timed step's launches)")
    ap.add_argument("--bytes", type=int, default=GIB, help="stream: bytes per GPU (default 1 GiB, the BASELINE config); host-abi: bytes per call (default 256 MiB)")
    ap.add_argument("--buffers", type=int, default=65536, help="cfg4: buffers in the batch")
    ap.add_argument("--buffer-bytes", type=int, default=16384, help="cfg4: bytes per buffer")
    ap.add_argument("--cpu-sample-mib", type=int, default=96, help="prefix of the stream timed on the CPU oracle")
    ap.add_argument("--header-items", type=int, default=1 << 20, help="items of the header_items leg (16..80 bytes each)")
    ap.add_argument("--stage-events-every", type=int, default=4,
                    help="stream: HIP events between the kernels in every Nth timed step only (the medians are of those steps)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="stream, one GPU: leave out the short cfg4 and host-abi legs that ride along as extra keys")
    ap.add_argument("--library", default=None,
                    help="shared library to load instead of the HIP build (tests/test_multi_rank.py passes the "
                         "CPU emulator build to exercise the rank logic where there is no GPU)")
    return ap.parse_args()


class Ranks:
    """World of one, or torch.distributed over gloo (timing exchange only, CPU tensors)."""

    def __init__(self, want):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        if self.world > 1:
            import torch
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world)
            self.dist, self.torch = dist, torch
        if want != self.world and self.rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d; running %d rank(s)" % (want, self.world, self.world),
                  file=sys.stderr)

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def _reduce(self, x, op):
        if not self.dist:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=op)
        return float(t[0])

    def max(self, x):
        return self._reduce(x, self.dist.ReduceOp.MAX if self.dist else None)

    def sum(self, x):
        return self._reduce(x, self.dist.ReduceOp.SUM if self.dist else None)

    def gather(self, obj):
        """obj of every rank, in rank order, on every rank."""
        if not self.dist:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def close(self):
        if self.dist:
            self.dist.destroy_process_group()


# ----------------------------------------------------------------------------- CPU baseline

def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores():
    """Cores this process may actually run on: the scheduler affinity mask, cut down by a cgroup CPU quota when there
    is one (os.cpu_count() is the machine's: on a leased box it said 256 where 12.8 cores' worth of work got done)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            fields = open(path).read().split()
            if path.endswith("cpu.max"):
                if fields[0] != "max":
                    quota = int(fields[0]) / int(fields[1])
            else:
                q = int(fields[0])
                if q > 0:
                    quota = q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError, ZeroDivisionError):
            continue
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def cpu_baseline(sample_bytes, seed, batch_seconds=8.0, buffer_bytes=16384):
    """The oracle on the host: the stream on one core (bounded prefix), the configs[3] batch on all cores."""
    import numpy as np

    import harness

    oracle = harness.oracle_codec()
    patterns, lens = harness.load_table()
    coder = oracle.lib.oracle_table_coder_new(patterns, lens)
    data = harness.splitmix64_bytes(seed, sample_bytes)
    dst = np.zeros(sample_bytes * 2 + 64, dtype=np.uint8)
    enc = oracle.new_encoder(coder)
    t0 = time.perf_counter()
    r = oracle.encode_call(enc, data, 0, dst, 0, dst.size)
    t1 = time.perf_counter()
    assert r.rc == 0
    back = np.zeros(sample_bytes, dtype=np.uint8)
    dec = oracle.new_decoder(coder)
    t2 = time.perf_counter()
    r2 = oracle.decode_call(dec, dst, 0, r.produced, back, 0, sample_bytes)
    t3 = time.perf_counter()
    assert r2.rc == 0 and np.array_equal(back, data)
    t_enc, t_dec = t1 - t0, t3 - t2

    # configs[3] on all host cores: one thread per core, each encoding + decoding 16 KiB buffers of its own for a
    # bounded time -- pthreads inside the oracle library (Python threads spent the time on the interpreter lock:
    # 0.24 GiB/s on 256 cores where this gives the cores' real rate)
    cores, quota = usable_cores()  # one thread per core this process can USE (affinity mask and cgroup quota)
    oracle.lib.oracle_batch_round_trips.restype = C.c_uint64
    oracle.lib.oracle_batch_round_trips.argtypes = [C.c_void_p, C.c_uint32, C.c_double, C.c_uint32, C.POINTER(C.c_double)]
    took = C.c_double()
    n_done = oracle.lib.oracle_batch_round_trips(coder, cores, batch_seconds, buffer_bytes, C.byref(took))
    assert n_done > 0, "the oracle's batch did not round-trip"
    done, tb = [n_done], took.value
    return {
        "value": round(sample_bytes / GIB / (t_enc + t_dec), 5),
        "unit": "GiB/s",
        "cores": 1,
        "kind": "port",
        "calibration": "none against the real reference: source/huffman.c needs aws-c-common's headers, which this image "
                       "lacks (stand-ins are not allowed for a reference build), so oracle/_ref holds only the "
                       "reference's generator tool.  The port (oracle/huffman_oracle.c) restates source/huffman.c line "
                       "by line -- one callback per symbol, one byte store per output byte -- and is pinned to the "
                       "reference's own vectors (tests/test_oracle_pins.py)",
        "sample": "first %d MiB of the same splitmix64 stream, encode %.2f s + decode %.2f s, C oracle -O2 "
                  "(scalar per-symbol callbacks like reference source/huffman.c), one core: the reference is "
                  "single-threaded per stream" % (sample_bytes >> 20, t_enc, t_dec),
        "encode_GiBps": round(sample_bytes / GIB / t_enc, 5),
        "decode_GiBps": round(sample_bytes / GIB / t_dec, 5),
        "cpu_model": cpu_model(),
        "host_cores": cores,
        "host_cores_how": "len(os.sched_getaffinity(0)) = %d, cgroup cpu quota = %s, os.cpu_count() = %s" % (
            len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else -1,
            "none" if quota is None else "%.1f" % quota, os.cpu_count()),
        "all_cores_cfg4": {
            "value": round(sum(done) * buffer_bytes / GIB / tb, 5),
            "unit": "GiB/s",
            "cores": cores,
            "sample": "%d x %d B buffers (BASELINE configs[3] shape) encoded + decoded in %.1f s by %d threads, one per "
                      "usable host core" % (sum(done), buffer_bytes, tb, cores),
        },
    }


# ----------------------------------------------------------------------------- timing helpers

class Stages:
    """HIP events between the kernels of every timed step; medians per stage afterwards."""

    def __init__(self, eng, steps, names_e, names_d, plans_e=1):
        self.eng, self.steps = eng, steps
        self.names_e, self.names_d = names_e, names_d
        self.ev_e = [[eng.new_events(4) for _ in range(plans_e)] for _ in range(steps)]
        self.ev_d = [eng.new_events(4) for _ in range(steps)]
        self.timed = list(range(steps))  # the steps whose launches carry events

    def medians(self):
        per_step = {k: [] for k in self.names_e + self.names_d}
        t_enc, t_dec = [], []
        for k in self.timed:
            e_sum = 0.0
            for i, name in enumerate(self.names_e):
                ms = sum(self.eng.elapsed_ms(ev[i], ev[i + 1]) for ev in self.ev_e[k])
                per_step[name].append(ms)
                e_sum += ms
            d_sum = 0.0
            for i, name in enumerate(self.names_d):
                ms = self.eng.elapsed_ms(self.ev_d[k][i], self.ev_d[k][i + 1])
                per_step[name].append(ms)
                d_sum += ms
            t_enc.append(e_sum)
            t_dec.append(d_sum)
        kernel_ms = {k: statistics.median(v) for k, v in per_step.items()}
        self.per_step = per_step
        return kernel_ms, statistics.median(t_enc), statistics.median(t_dec)

    def gave_up_steps(self):
        """Timed steps in which the one-pass encoder gave up and the kernel queued behind it did the launch over (a silent
        demotion otherwise: the output is right either way).  Told from the stage that holds both: the way back, taken,
        is five times the kernel it is the way back of (tiles by ticket: ~3 ms a GiB); not taken, an empty launch."""
        if self.names_e[0] != "enc_onepass":
            return 0
        first = self.per_step[self.names_e[0]]
        floor = statistics.median(first)
        return sum(1 for ms in first if ms > 3 * floor)


def roofline_of(label, kernels, algo_bytes, ms, traffic_table):
    achieved = algo_bytes / max(ms * 1e-3, 1e-12)
    traffic = None
    if traffic_table and all(k in traffic_table for k in kernels):
        parts = [traffic_table[k].get("hbm_bytes_per_launch") for k in kernels]
        traffic = sum(parts) if all(p is not None for p in parts) else None
    elif traffic_table and any(k in traffic_table for k in kernels):
        # (a stage without an entry of its own moves nothing worth a counter pass: a few result records)
        parts = [traffic_table[k].get("hbm_bytes_per_launch") for k in kernels if k in traffic_table]
        traffic = sum(parts) if all(p is not None for p in parts) else None
    return {
        "path": label,
        "kernels": kernels,
        "bound": "hbm",
        "achieved": round(achieved / 1e9, 2),
        "peak": HBM_PEAK_BYTES_PER_S / 1e9,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_BYTES_PER_S, 4),
        "traffic": traffic,
        "algorithmic_bytes_per_launch": algo_bytes,
        "ms": round(ms, 4),
    }


def load_traffic(name="pmc_traffic.json"):
    path = os.path.join(REPO, "profiles", name)
    return json.load(open(path)) if os.path.exists(path) else None


def stage_names(lib, eng):
    one_pass = bool(lib.aws_huffman_amd_engine_encodes_in_one_pass(eng.h))
    names_e = ["enc_onepass", "enc_finish", "enc_tiny"] if one_pass else ["enc_count", "enc_scan", "enc_pack"]
    return names_e, ["dec_sync", "dec_scan", "dec_emit"]


def digest_of(eng, ptr, size):
    h = hashlib.sha256()
    for off in range(0, size, 256 << 20):
        h.update(eng.download(ptr, min(256 << 20, size - off), offset=off).tobytes())
    return h.hexdigest()


def rooflines(out, names_e, names_d, kernel_ms, t_enc_ms, t_dec_ms, n, e_len, algo_per_kernel, traffic_measured=True, traffic_file="pmc_traffic.json"):
    # (the committed counter passes are per workload: the 1 GiB stream's, BASELINE configs[3]'s; any other workload's
    # `traffic` is null, not another workload's figure)
    traffic = load_traffic(traffic_file) if traffic_measured else None
    traffic_measured = traffic is not None
    enc = roofline_of("encode", names_e, n + e_len, t_enc_ms, traffic)
    dec = roofline_of("decode", names_d, n + e_len, t_dec_ms, traffic)
    out["roofline_encode"], out["roofline_decode"] = enc, dec
    out["roofline"] = dec if t_dec_ms >= t_enc_ms else enc
    out["traffic_source"] = "none: HBM traffic was not measured for this workload" if not traffic_measured else (traffic or {}).get("_source", "profiles/pmc_traffic.json: HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, "
                                                "separate rocprofv3 --pmc passes of this command, committed profile), not this run")
    dominant = max(kernel_ms, key=lambda k: kernel_ms[k])
    out["roofline_kernel"] = roofline_of(dominant, [dominant], algo_per_kernel[dominant], kernel_ms[dominant], traffic)
    out["encode_read_frac_of_hbm_peak"] = round(n / max(t_enc_ms * 1e-3, 1e-12) / HBM_PEAK_BYTES_PER_S, 4)


# ----------------------------------------------------------------------------- workloads

def run_stream(args, ranks, lib, eng):
    import harness

    n = args.bytes
    seed = 5 + ranks.rank  # cfg5 = seed 5 + g
    worst = n * 10 // 8 + 64
    d_in, d_enc, d_back = eng.alloc(n), eng.alloc(worst), eng.alloc(n + 64)
    eng.fill_splitmix64(d_in, n, seed)
    enc_plan = eng.encode_plan([dict(in_offset=0, in_len=n, out_offset=0, out_capacity=worst)])

    # untimed first pass: learn the encoded length, build the decode plan, check both record sets.  The digests of what
    # the passes wrote are taken AFTER the timed steps (hashing three GiB on the host leaves the GPU idle for seconds, and
    # its clocks need ~13 steps to settle after that -- profiles/tools/step_ramp.py: the set-up would put that ramp into
    # the timed steps); a stream that differs from the pinned one still ends the run without a line.
    eng.encode_launch(enc_plan, d_in, d_enc)
    (rc, err, consumed, e_len, _, _), = eng.encode_results(enc_plan, 1)
    assert rc == 0 and consumed == n, (rc, err, consumed)
    dec_plan = eng.decode_plan([dict(in_offset=0, in_len=e_len, out_offset=0, out_capacity=n)])
    eng.decode_launch(dec_plan, d_enc, d_back)
    (rc, err, symbols, _), = eng.decode_results(dec_plan, 1)
    assert rc == 0 and symbols == n, (rc, err, symbols)
    pinned, pinned_by = None, None
    if n == GIB and seed == 5:
        pinned = harness.load_json("survey_probe_records.json")["streams"]["G1G"]
        pinned_by = "the survey's record of the real reference (tests/golden/survey_probe_records.json)"
    elif n == GIB and str(seed) in harness.load_json("config5_stream_pins.json")["streams"]:
        # configs[4]: rank g's stream is seed 5 + g; seeds 6 .. 12 are pinned by the oracle (tests/golden/make_config5_pins.py)
        pinned = harness.load_json("config5_stream_pins.json")["streams"][str(seed)]
        pinned_by = "the pinned oracle's record (tests/golden/config5_stream_pins.json)"
    if pinned:
        assert e_len == pinned["encoded_len"], "rank %d: encoded length differs from the pinned one" % ranks.rank

    names_e, names_d = stage_names(lib, eng)
    stages = Stages(eng, args.steps, names_e, names_d)

    every = max(1, min(args.stage_events_every, args.steps))
    stages.timed = [k for k in range(args.steps) if k % every == 0]

    def step(k=None):
        with_events = k is not None and k % every == 0
        eng.encode_launch(enc_plan, d_in, d_enc, events=stages.ev_e[k][0] if with_events else None)
        eng.decode_launch(dec_plan, d_enc, d_back, events=stages.ev_d[k] if with_events else None)

    for _ in range(args.warmup):
        step()
    eng.sync()
    ranks.barrier()
    eng.sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    eng.sync()
    ranks.barrier()
    wall = ranks.max(time.perf_counter() - t0)

    # what the TIMED steps left behind: both record sets, both digests, the roads taken (untimed)
    (rc, err, consumed, e_after, _, _), = eng.encode_results(enc_plan, 1)
    (rc2, err2, symbols, _), = eng.decode_results(dec_plan, 1)
    in_digest, enc_digest = digest_of(eng, d_in, n), digest_of(eng, d_enc, e_len)
    if pinned:
        assert in_digest == pinned.get("sha256_input", in_digest), "rank %d: input stream differs from the pinned one" % ranks.rank
        assert enc_digest == pinned["sha256_encoded"], "rank %d: encoded stream differs from the pinned one" % ranks.rank
    bit_exact = ((rc, consumed, e_after) == (0, n, e_len) and (rc2, symbols) == (0, n) and digest_of(eng, d_back, n) == in_digest)
    assert bit_exact, "the timed steps did not leave the stream's round trip"
    roads = {0: "two-pass", 1: "one-pass", 2: "one-pass gave up, two-pass did the launch over"}

    kernel_ms, t_enc_ms, t_dec_ms = stages.medians()
    out = {
        "config": {
            "workload": "1 GiB splitmix64 bytes per GPU (seed 5+rank), test_huffman_static_table coder: "
                        "encode (BASELINE configs[1]) then decode back (configs[2]), HBM-resident",
            "stream_bytes": n, "encoded_bytes": e_len, "bit_exact": bit_exact, "sha256_encoded": enc_digest,
            "bit_exact_checked": "records and sha256 of the encoded stream and of the decoded bytes after the last timed step",
            "encode_road": roads[eng.encode_road(enc_plan)].replace("two-pass", "three-kernel"),
            "decode_road": roads[eng.decode_road(dec_plan)],
            # timed steps in which the one-pass encoder gave up (a wait ran out: something else held CUs) and the
            # kernel queued behind it did the launch over on the device -- right output, slower step
            "gave_up": stages.gave_up_steps(),
        },
        "scaling": "weak",
    }
    algo = {names_e[0]: n + e_len if names_e[0] == "enc_onepass" else n, names_e[1]: 0, names_e[2]: 0 if names_e[0] == "enc_onepass" else n + e_len,
            "dec_sync": e_len, "dec_scan": 0, "dec_emit": e_len + n}
    rooflines(out, names_e, names_d, kernel_ms, t_enc_ms, t_dec_ms, n, e_len, algo)
    per_rank = {"rank": ranks.rank, "seed": seed, "encoded_bytes": e_len, "sha256_encoded": enc_digest, "pinned_by": pinned_by,
                "encode_ms": round(t_enc_ms, 4), "decode_ms": round(t_dec_ms, 4)}
    return out, n, e_len, wall, kernel_ms, t_enc_ms, t_dec_ms, per_rank, seed


def run_header_items(lib, eng, n_items=1 << 20, steps=5, warmup=2):
    """A million header-field-sized strings (16..80 printable bytes each, 48 MiB: what the reference's one consumer, HPACK,
    hands over one call at a time), as ONE batch: encode launch -> the decode plan made on the device from that launch's
    records (aws_huffman_amd_decode_plan_from_encode: the encoded lengths never come to the host) -> decode launch ->
    synchronise.  Wall clock per such round trip; the decoded bytes are compared with the input."""
    import numpy as np

    import harness

    rng = np.random.default_rng(11)
    lens = rng.integers(16, 81, n_items)
    in_offs = np.concatenate([[0], np.cumsum(lens[:-1])]).astype(np.int64)
    total_in, cap = int(lens.sum()), 128
    arr = (harness.AmdEncodeItem * n_items)()
    for i in range(n_items):
        arr[i].in_offset, arr[i].in_len, arr[i].out_offset, arr[i].out_capacity = int(in_offs[i]), int(lens[i]), i * cap, cap
        arr[i].eos_padding = 0xFF
    d_in, d_enc, d_back = eng.alloc(total_in + 64), eng.alloc(n_items * cap + 64), eng.alloc(total_in + 64)
    data = (32 + harness.splitmix64_bytes(17, total_in) % 95).astype(np.uint8)
    eng.upload(d_in, data)
    eplan, dplan = C.c_void_p(), C.c_void_p()
    t0 = time.perf_counter()
    assert lib.aws_huffman_amd_encode_plan_new(C.byref(eplan), eng.h, arr, n_items) == 0
    plan_ms = (time.perf_counter() - t0) * 1e3
    assert lib.aws_huffman_amd_decode_plan_new(C.byref(dplan), eng.h, None, 0) == 0
    lib.aws_huffman_amd_decode_plan_from_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    # the same records lying in device memory (a caller that makes them there, or keeps them): the plan without a host loop
    d_items = eng.alloc(C.sizeof(arr))
    eng.upload(d_items, np.frombuffer(arr, dtype=np.uint8))
    fresh_plan = eng.empty_encode_plan()
    dev_plan_ms = []
    for _ in range(3):
        t0 = time.perf_counter()
        assert lib.aws_huffman_amd_encode_plan_reset_device_items(fresh_plan, d_items, n_items, None) == 0
        dev_plan_ms.append((time.perf_counter() - t0) * 1e3)

    def fresh_round_trip():
        # nothing kept from a batch before: encode plan from the records, launch, chained decode plan, launch, wait
        assert lib.aws_huffman_amd_encode_plan_reset_device_items(fresh_plan, d_items, n_items, None) == 0
        eng.encode_launch(fresh_plan, d_in, d_enc)
        assert lib.aws_huffman_amd_decode_plan_from_encode(dplan, fresh_plan, None) == 0
        eng.decode_launch(dplan, d_enc, d_back)
        eng.sync()

    fresh = []
    for k in range(warmup + steps):
        t0 = time.perf_counter()
        fresh_round_trip()
        fresh.append((time.perf_counter() - t0) * 1e3)
    fresh_ms = statistics.median(fresh[warmup:])

    def round_trip():
        eng.encode_launch(eplan, d_in, d_enc)
        assert lib.aws_huffman_amd_decode_plan_from_encode(dplan, eplan, None) == 0
        eng.decode_launch(dplan, d_enc, d_back)
        eng.sync()

    times = []
    for k in range(warmup + steps):
        if k == warmup:
            eng.fill(d_back, 0xEE, total_in)  # (the timed steps decode into scrambled memory)
        t0 = time.perf_counter()
        round_trip()
        times.append((time.perf_counter() - t0) * 1e3)
    ms = statistics.median(times[warmup:])
    eres, dres = eng.encode_results(eplan, n_items), eng.decode_results(dplan, n_items)
    bit_exact = (all(r[0] == 0 for r in eres) and all(r[0] == 0 and r[2] == int(lens[i]) for i, r in enumerate(dres)) and
                 np.array_equal(eng.download(d_back, total_in), data))
    assert bit_exact, "the batch of header-sized items did not round-trip"
    lib.aws_huffman_amd_encode_plan_destroy(eplan)
    lib.aws_huffman_amd_encode_plan_destroy(fresh_plan)
    lib.aws_huffman_amd_decode_plan_destroy(dplan)
    for ptr in (d_in, d_enc, d_back, d_items):
        eng.free(ptr)
    return {"workload": "%d items of 16..80 printable bytes (%.1f MiB), one batch: encode launch, decode plan chained to it on the "
                        "device (no lengths on the host), decode launch, synchronise" % (n_items, total_in / (1 << 20)),
            "bit_exact": bit_exact, "steps": steps, "round_trip_ms": round(ms, 4), "items_per_s": round(n_items / (ms * 1e-3)),
            "value_GiBps": round(total_in / GIB / (ms * 1e-3), 2),
            # what a batch that is NEW costs: its encode plan made from the records in device memory (three small launches and
            # a wait for a few totals), launch, chained decode plan, launch, wait -- nothing kept from a batch before
            "fresh_round_trip_ms": round(fresh_ms, 4), "fresh_items_per_s": round(n_items / (fresh_ms * 1e-3)),
            "plan_ms": {"encode": round(statistics.median(dev_plan_ms), 3), "encode_from_host_records": round(plan_ms, 3),
                        "decode_chained_call": "queued with the launches"},
            "timing": "host wall clock around the three calls and the synchronise, median of %d" % steps}


def run_cfg4(args, ranks, lib, eng):
    import numpy as np

    import harness

    count_all, size = args.buffers, args.buffer_bytes
    mine = list(range(ranks.rank, count_all, ranks.world))  # buffer i -> rank i mod N
    count = len(mine)
    stride = 2 * size
    d_in, d_out, d_back = eng.alloc(count * size), eng.alloc(count * stride), eng.alloc(count * size)
    for k, i in enumerate(mine):
        assert lib.aws_huffman_amd_device_fill_splitmix64(eng.h, d_in + k * size, size, 2 + i) == 0
    eng.fill(d_out, 0x5A, count * stride)
    # every fourth buffer of the BATCH has room for 16 384 bytes only: SHORT_BUFFER, then a second call
    items = [dict(in_offset=k * size, in_len=size, out_offset=k * stride, out_capacity=size if i % 4 == 0 else stride)
             for k, i in enumerate(mine)]
    plan = eng.encode_plan(items)
    plan_ms = {"encode": round(eng.last_plan_ms, 3)}
    eng.encode_launch(plan, d_in, d_out)
    res = eng.encode_results(plan, count)
    short = [k for k, i in enumerate(mine) if i % 4 == 0]
    assert all(res[k][0] == -1 and res[k][1] == harness.AWS_ERROR_SHORT_BUFFER and res[k][3] == size for k in short)
    assert all(res[k][0] == 0 and res[k][2] == size for k in range(count) if k not in set(short))
    if mine and mine[0] == 0 and size == 16384:
        first = harness.load_json("survey_probe_records.json")["G16K_partial_encode"][2]
        assert res[0] == (-1, harness.AWS_ERROR_SHORT_BUFFER, first["consumed"], first["out_len"],
                          first["overflow_num_bits"], first["overflow_pattern"]), "buffer 0 differs from the reference's record"
    resume = [dict(in_offset=k * size + res[k][2], in_len=size - res[k][2], out_offset=k * stride + size,
                   out_capacity=size, overflow_in=(res[k][5], res[k][4])) for k in short]
    plan2 = eng.encode_plan(resume) if resume else None
    plan_ms["encode_resume"] = round(eng.last_plan_ms, 3) if resume else 0.0
    lengths = [r[3] for r in res]
    if plan2:
        eng.encode_launch(plan2, d_in, d_out)
        res2 = eng.encode_results(plan2, len(resume))
        assert all(r[0] == 0 for r in res2)
        for k, r in zip(short, res2):
            lengths[k] += r[3]
    dplan = eng.decode_plan([dict(in_offset=k * stride, in_len=lengths[k], out_offset=k * size, out_capacity=size)
                             for k in range(count)])
    plan_ms["decode"] = round(eng.last_plan_ms, 3)
    eng.decode_launch(dplan, d_out, d_back)
    dres = eng.decode_results(dplan, count)
    assert all(r[0] == 0 and r[2] == size for r in dres)
    assert digest_of(eng, d_back, count * size) == digest_of(eng, d_in, count * size), "round trip is not bit-exact"
    h = hashlib.sha256()
    enc_all = eng.download(d_out, count * stride)
    for k in range(count):
        h.update(enc_all[k * stride:k * stride + lengths[k]].tobytes())
    enc_digest = h.hexdigest()
    n, e_len = count * size, int(sum(lengths))

    names_e, names_d = stage_names(lib, eng)
    stages = Stages(eng, args.steps, names_e, names_d, plans_e=2 if plan2 else 1)

    def step(k=None):
        eng.encode_launch(plan, d_in, d_out, events=stages.ev_e[k][0] if k is not None else None)
        if plan2:
            eng.encode_launch(plan2, d_in, d_out, events=stages.ev_e[k][1] if k is not None else None)
        eng.decode_launch(dplan, d_out, d_back, events=stages.ev_d[k] if k is not None else None)

    for _ in range(args.warmup):
        step()
    eng.sync()
    ranks.barrier()
    eng.sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    eng.sync()
    ranks.barrier()
    wall = ranks.max(time.perf_counter() - t0)
    dres = eng.decode_results(dplan, count)
    bit_exact = all(r[0] == 0 and r[2] == size for r in dres) and digest_of(eng, d_back, count * size) == digest_of(eng, d_in, count * size)
    assert bit_exact, "the timed steps did not leave the batch's round trip"
    kernel_ms, t_enc_ms, t_dec_ms = stages.medians()

    fresh_keys = {}
    if not getattr(args, "no_fresh", False):
        # ---- a batch that is NEW: every plan made for it, from records in device memory (no loop over the items on the host),
        # with what the reference's contract makes the host do in between -- look at the records of the first call to learn
        # which buffers ran out of room and with what carried bits, then call again for those.
        enc_arr = eng._encode_item_array(items)
        d_items1 = eng.alloc(C.sizeof(enc_arr))
        eng.upload(d_items1, np.frombuffer(enc_arr, dtype=np.uint8))
        res_arr = eng._encode_item_array(resume) if resume else None
        d_items2 = eng.alloc(C.sizeof(res_arr)) if resume else None
        dec_arr = eng._decode_item_array([dict(in_offset=k * stride, in_len=lengths[k], out_offset=k * size, out_capacity=size) for k in range(count)])
        d_items3 = eng.alloc(C.sizeof(dec_arr))
        f1, f2, f3 = eng.empty_encode_plan(), eng.empty_encode_plan(), eng.empty_decode_plan()
        fresh_plan_ms = {"encode": [], "encode_resume": [], "decode": []}
        fresh_ms = []
        res_buf = (harness.AmdEncodeResult * count)()
        for rep_k in range(4):
            t_start = time.perf_counter()
            t0 = time.perf_counter()
            assert lib.aws_huffman_amd_encode_plan_reset_device_items(f1, d_items1, count, None) == 0
            fresh_plan_ms["encode"].append((time.perf_counter() - t0) * 1e3)
            eng.encode_launch(f1, d_in, d_out)
            assert lib.aws_huffman_amd_encode_plan_results(f1, res_buf, None) == 0  # (the records: which buffers want a second call)
            if resume:
                t0 = time.perf_counter()
                eng.upload(d_items2, np.frombuffer(res_arr, dtype=np.uint8))
                assert lib.aws_huffman_amd_encode_plan_reset_device_items(f2, d_items2, len(resume), None) == 0
                fresh_plan_ms["encode_resume"].append((time.perf_counter() - t0) * 1e3)
                eng.encode_launch(f2, d_in, d_out)
                assert lib.aws_huffman_amd_encode_plan_results(f2, res_buf, None) == 0
            t0 = time.perf_counter()
            eng.upload(d_items3, np.frombuffer(dec_arr, dtype=np.uint8))  # (the streams' lengths: known on the host from the records)
            assert lib.aws_huffman_amd_decode_plan_reset_device_items(f3, d_items3, count, None) == 0
            fresh_plan_ms["decode"].append((time.perf_counter() - t0) * 1e3)
            eng.decode_launch(f3, d_out, d_back)
            eng.sync()
            fresh_ms.append((time.perf_counter() - t_start) * 1e3)
        fres = eng.decode_results(f3, count)
        assert all(r[0] == 0 and r[2] == size for r in fres) and digest_of(eng, d_back, count * size) == digest_of(eng, d_in, count * size)
        # ... and with room for every output: encode plan from a STRIDE, decode plan chained to the launch on the device -- not one
        # record comes to the host before the end
        roomy = eng.empty_encode_plan()
        chained = eng.empty_decode_plan()
        desc = harness.StridedItems(count=count, in_offset=0, in_stride=size, in_len=size, out_offset=0, out_stride=stride,
                                    out_capacity=stride, eos_padding=0xFF)
        lib.aws_huffman_amd_decode_plan_from_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        chained_ms, chained_plan_ms = [], {"encode": [], "decode": []}
        for rep_k in range(4):
            eng.fill(d_back, 0xEE, count * size)
            eng.sync()
            t_start = time.perf_counter()
            assert lib.aws_huffman_amd_encode_plan_reset_strided(roomy, C.byref(desc), None) == 0
            chained_plan_ms["encode"].append((time.perf_counter() - t_start) * 1e3)
            eng.encode_launch(roomy, d_in, d_out)
            t0 = time.perf_counter()
            assert lib.aws_huffman_amd_decode_plan_from_encode(chained, roomy, None) == 0
            chained_plan_ms["decode"].append((time.perf_counter() - t0) * 1e3)  # (includes the wait for the encode launch)
            eng.decode_launch(chained, d_out, d_back)
            eng.sync()
            chained_ms.append((time.perf_counter() - t_start) * 1e3)
        cres = eng.decode_results(chained, count)
        assert all(r[0] == 0 and r[2] == size for r in cres) and digest_of(eng, d_back, count * size) == digest_of(eng, d_in, count * size)
        for pl in (f1, f2, roomy):
            lib.aws_huffman_amd_encode_plan_destroy(pl)
        for pl in (f3, chained):
            lib.aws_huffman_amd_decode_plan_destroy(pl)
        for ptr in (d_items1, d_items2, d_items3):
            if ptr:
                eng.free(ptr)
        med = statistics.median
        fresh_keys = {
            "fresh_batch_ms": round(med(fresh_ms[1:]), 4),
            "fresh_batch_GiBps": round(count * size / GIB / (med(fresh_ms[1:]) * 1e-3), 2),
            "fresh_plan_ms": {k: round(med(v[1:]), 3) if v else 0.0 for k, v in fresh_plan_ms.items()},
            "fresh_chained_ms": round(med(chained_ms[1:]), 4),
            "fresh_chained_GiBps": round(count * size / GIB / (med(chained_ms[1:]) * 1e-3), 2),
            "fresh_chained_plan_ms": {k: round(med(v[1:]), 3) for k, v in chained_plan_ms.items()},
            "fresh": "host wall clock, median of 3: every plan made for the batch from records in device memory (fresh_batch: with "
                     "the reference's SHORT_BUFFER -> second call in between, records fetched; fresh_chained: room for every "
                     "output, encode plan from a stride, decode plan chained on the device), launches, waits",
        }
    out = {
        "config": {
            "workload": "BASELINE configs[3]: %d buffers x %d B (buffer i = splitmix64 seed 2+i, rank r takes i = r mod N), "
                        "every fourth one capacity-limited (SHORT_BUFFER record, then a resume call), all decoded back, "
                        "HBM-resident" % (count_all, size),
            "buffers": count_all, "buffer_bytes": size, "bit_exact": bit_exact,
            # what MAKING the three plans of a step cost on this host (aws_huffman_amd_*_plan_new, device allocations
            # included; a decode plan depends on the encoded lengths, so a fresh batch pays it): not in ms_per_step,
            # which times launches of plans that exist
            "plan_ms": plan_ms,
            "gave_up": stages.gave_up_steps(),
        },
        "scaling": "strong",
    }
    out["config"].update(fresh_keys)
    algo = {names_e[0]: n + e_len if names_e[0] == "enc_onepass" else n, names_e[1]: 0, names_e[2]: 0 if names_e[0] == "enc_onepass" else n + e_len,
            "dec_sync": e_len, "dec_scan": 0, "dec_emit": e_len + n}
    # (HBM traffic: counter passes of this workload at 16 KiB a buffer and the full batch, profiles/tools/cfg4_traffic.sh)
    rooflines(out, names_e, names_d, kernel_ms, t_enc_ms, t_dec_ms, n, e_len, algo,
              traffic_measured=(size == 16384 and count_all == 65536 and ranks.world == 1), traffic_file="pmc_traffic_cfg4.json")
    per_rank = {"rank": ranks.rank, "buffers": count, "encoded_bytes": e_len, "sha256_encoded_streams": enc_digest,
                "encode_ms": round(t_enc_ms, 4), "decode_ms": round(t_dec_ms, 4)}
    return out, n, e_len, wall, kernel_ms, t_enc_ms, t_dec_ms, per_rank, 2


def run_host_abi(args, ranks, lib, eng, coder):
    """aws_huffman_encode / aws_huffman_decode on host memory: what a caller of the reference's API gets."""
    import numpy as np

    import harness

    n = min(args.bytes, 256 << 20) if args.workload == "stream" else (args.bytes if args.bytes != GIB else 256 << 20)
    seed = 5 + ranks.rank
    codec = harness.Codec(lib, "aws_")
    data = harness.splitmix64_bytes(seed, n)
    dst = np.zeros(n * 10 // 8 + 64, dtype=np.uint8)
    back = np.zeros(n, dtype=np.uint8)

    def step():
        enc = codec.new_encoder(coder)
        t0 = time.perf_counter()
        r = codec.encode_call(enc, data, 0, dst, 0, dst.size)
        t1 = time.perf_counter()
        dec = codec.new_decoder(coder)
        r2 = codec.decode_call(dec, dst, 0, r.produced, back, 0, n)
        t2 = time.perf_counter()
        assert r.rc == 0 and r.consumed == n and r2.rc == 0 and r2.produced == n
        return r.produced, (t1 - t0) * 1e3, (t2 - t1) * 1e3

    e_len = 0
    for _ in range(max(args.warmup, 1)):
        e_len, _, _ = step()
    assert np.array_equal(back, data), "round trip is not bit-exact"
    ranks.barrier()
    t0 = time.perf_counter()
    enc_ms, dec_ms = [], []
    for _ in range(args.steps):
        _, a, b = step()
        enc_ms.append(a)
        dec_ms.append(b)
    ranks.barrier()
    wall = ranks.max(time.perf_counter() - t0)
    t_enc_ms, t_dec_ms = statistics.median(enc_ms), statistics.median(dec_ms)
    link = (n + e_len) / max(min(t_enc_ms, t_dec_ms) * 1e-3, 1e-12)
    out = {
        "config": {
            "workload": "aws_huffman_encode + aws_huffman_decode of one %d MiB buffer in pageable HOST memory (the "
                        "reference's own entry points): H2D + kernels + D2H per call" % (n >> 20),
            "stream_bytes": n, "encoded_bytes": e_len, "bit_exact": True,
        },
        "scaling": "weak",
        "roofline": {"path": "host link", "bound": "pcie", "achieved": round(link / 1e9, 2),
                     "peak": PCIE_PEAK_BYTES_PER_S / 1e9, "unit": "GB/s", "frac": round(link / PCIE_PEAK_BYTES_PER_S, 4),
                     "traffic": None, "note": "bytes over the host link (N + E) / the faster of the two calls; not an HBM figure"},
    }
    kernel_ms = {"aws_huffman_encode": t_enc_ms, "aws_huffman_decode": t_dec_ms}
    per_rank = {"rank": ranks.rank, "seed": seed, "encoded_bytes": e_len,
                "encode_ms": round(t_enc_ms, 4), "decode_ms": round(t_dec_ms, 4)}
    return out, n, e_len, wall, kernel_ms, t_enc_ms, t_dec_ms, per_rank, seed


def spawn_ranks(n):
    """`python bench.py --gpus N` with no launcher around it: this process becomes the launcher.  N fresh children, one
    per GPU, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as torch.distributed.run would -- started before this
    process has loaded the HIP library or touched a device (a process that has initialised the GPU must not exec or
    fork workers on this pool), each its own stream and device, nothing shared but the gloo barrier and the
    max-reduction of the step time.  Rank 0's stdout is this process's: its one JSON line."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    children = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.setdefault("OMP_NUM_THREADS", "1")
        children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                         stdout=None if rank == 0 else subprocess.DEVNULL))
    codes = []
    try:
        for child in children:
            codes.append(child.wait())
    finally:
        for child in children:
            if child.poll() is None:
                child.kill()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        raise SystemExit("bench.py: rank(s) failed: %r" % bad)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)
    import harness  # the HIP library is loaded before anything else (torch, in Ranks) can pull in another HIP runtime

    lib = harness.load_product(args.library)
    ranks = Ranks(args.gpus)
    if lib.aws_huffman_amd_device_count() < 1:
        raise SystemExit("bench.py: no HIP device visible and the product has no CPU path")
    patterns, lens = harness.load_table()
    coder = lib.aws_huffman_amd_table_coder_new(patterns, lens)
    ndev = lib.aws_huffman_amd_device_count()
    eng = harness.Engine(lib, coder, device=ranks.local_rank % ndev)

    if args.workload == "stream":
        out, n, e_len, wall, kernel_ms, t_enc_ms, t_dec_ms, per_rank, seed = run_stream(args, ranks, lib, eng)
    elif args.workload == "cfg4":
        out, n, e_len, wall, kernel_ms, t_enc_ms, t_dec_ms, per_rank, seed = run_cfg4(args, ranks, lib, eng)
    else:
        out, n, e_len, wall, kernel_ms, t_enc_ms, t_dec_ms, per_rank, seed = run_host_abi(args, ranks, lib, eng, coder)
    per_rank["device"] = ranks.local_rank % ndev

    ms_per_step = wall / args.steps * 1e3
    total_units = ranks.sum(float(n)) / GIB  # GiB of input symbols per step, all ranks
    line = {
        "metric": METRIC,
        "value": round(total_units / (wall / args.steps), 6),
        "unit": "GiB/s",
        "n_gpus": ranks.world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": out.pop("scaling"),
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "workload": args.workload,
    }
    line.update(out)
    line.update({
        "encode_GiBps": round(n / GIB / max(t_enc_ms * 1e-3, 1e-12), 2),
        "decode_GiBps_encoded_in": round(e_len / GIB / max(t_dec_ms * 1e-3, 1e-12), 2),
        "decode_GiBps_symbols_out": round(n / GIB / max(t_dec_ms * 1e-3, 1e-12), 2),
        "encode_path_frac_of_hbm_peak": round((n + e_len) / max(t_enc_ms * 1e-3, 1e-12) / HBM_PEAK_BYTES_PER_S, 4),
        "decode_path_frac_of_hbm_peak": round((n + e_len) / max(t_dec_ms * 1e-3, 1e-12) / HBM_PEAK_BYTES_PER_S, 4),
        "timing": "value and ms_per_step are the wall clock of all %d timed steps between barriers; kernel_ms and the path times "
                  "are medians over the timed steps whose launches carry HIP events between their kernels (on the engine's stream): "
                  "every %s of them -- the eight event records of a step are ~40 us of its 1.65 ms, and the steps without them "
                  "are the workload as a caller runs it" % (
                      args.steps, {1: "one", 2: "second", 3: "third", 4: "fourth"}.get(args.stage_events_every, "%dth" % args.stage_events_every)),
        "stage_events_every": args.stage_events_every,
        "ranks": ranks.gather(per_rank),
        "kernel_ms": {k: round(v, 4) for k, v in kernel_ms.items()},
    })
    if args.workload == "stream" and ranks.world == 1 and not args.no_extra_legs:
        # BASELINE configs[3] and the reference's own entry points on host memory ride along as extra keys (short legs,
        # never `value`): what the driver's default run would otherwise never measure
        def leg(fn, *more):
            o, ln, le, w, kms, te, td, _, _ = fn(*more)
            return {"workload": o["config"]["workload"], "bit_exact": o["config"]["bit_exact"], "steps": more[0].steps,
                    **({"plan_ms": o["config"]["plan_ms"]} if "plan_ms" in o["config"] else {}),
                    **{k: v for k, v in o["config"].items() if k.startswith("fresh")},
                    "value_GiBps": round(ln / GIB / max((te + td) * 1e-3, 1e-12), 2),
                    "encode_ms": round(te, 4), "decode_ms": round(td, 4),
                    "encode_path_frac_of_hbm_peak": round((ln + le) / max(te * 1e-3, 1e-12) / HBM_PEAK_BYTES_PER_S, 4),
                    "decode_path_frac_of_hbm_peak": round((ln + le) / max(td * 1e-3, 1e-12) / HBM_PEAK_BYTES_PER_S, 4),
                    "kernel_ms": {k: round(v, 4) for k, v in kms.items()}, "roofline": o.get("roofline")}
        short = argparse.Namespace(**vars(args))
        short.steps, short.warmup = 5, 2
        try:
            line["cfg4"] = leg(run_cfg4, short, ranks, lib, eng)
            # the same batch shape with 2 KiB items (128 MiB in 65 536 of them): between header size and a whole
            # segment / chunk, where a workgroup per item is mostly idle lanes ("mid-size items")
            mid = argparse.Namespace(**vars(short))
            mid.buffer_bytes = 2048
            line["mid_items"] = leg(run_cfg4, mid, ranks, lib, eng)
            line["mid_items"]["workload"] = line["mid_items"]["workload"].replace("BASELINE configs[3]", "configs[3]'s shape with 2 KiB items")
            line["header_items"] = run_header_items(lib, eng, n_items=args.header_items)
            short.steps, short.warmup = 3, 1  # (one 256 MiB buffer, or the stream's size when that was given and is smaller)
            line["host_abi"] = leg(run_host_abi, short, ranks, lib, eng, coder)
            line["host_abi"].pop("encode_path_frac_of_hbm_peak"), line["host_abi"].pop("decode_path_frac_of_hbm_peak")
        except MemoryError as exc:  # (a host without room for the legs: the headline line stands)
            line["extra_legs_error"] = repr(exc)
    if ranks.rank == 0 and ranks.world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(min(args.cpu_sample_mib << 20, n), seed)
    else:
        line["cpu_baseline"] = None
    if ranks.rank == 0:
        print(json.dumps(line), flush=True)
    ranks.close()


if __name__ == "__main__":
    main()
