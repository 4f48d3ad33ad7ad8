#!/usr/bin/env python3
"""Benchmark of the Huffman hot path on MI355X: encode + decode of 1 GiB of random bytes.

Contract (one JSON line on stdout from rank 0):

  python bench.py --gpus N --steps K --warmup W

  * workload = BASELINE.json configs[1]+[2]: one 1 GiB stream of splitmix64 bytes per GPU
    (seed 5 + rank), encoded with the test coder of the reference
    (tests/test_huffman_static_table.def) and decoded back.  One STEP = one encode of the
    stream + one decode of the result, inputs and outputs resident in HBM.
  * value = GiB of input symbols pushed through encode+decode per second, whole job:
    N_gpus * 2^30 B / (t_encode + t_decode) / 2^30, time = max over ranks between barriers.
  * roofline = the kernel with the largest share of the step, its algorithmic bytes
    (DESIGN.md "Kernels") over its average duration measured with HIP events recorded on the
    stream between the kernels of every timed step, against 8 TB/s of HBM.
  * cpu_baseline = the CPU oracle (a port of the reference's scalar loop; the reference itself
    needs aws-c-common and cannot be built here) timed on one host core on a bounded prefix.

For N > 1 the driver launches one process per GPU through torch.distributed.run; ranks share
nothing but a barrier and a max-reduction of the step time (no collective on the data path:
the streams are independent, DESIGN.md "Multi-GPU").
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "tests"))

GIB = 1 << 30
HBM_PEAK_BYTES_PER_S = 8.0e12  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bytes", type=int, default=GIB, help="stream length per GPU (default 1 GiB, the BASELINE config)")
    ap.add_argument("--cpu-sample-mib", type=int, default=96, help="prefix of the stream timed on the CPU oracle")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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

    def max(self, x):
        if not self.dist:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t[0])

    def sum(self, x):
        if not self.dist:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t[0])

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


def cpu_baseline(sample_bytes, seed):
    """Oracle encode + decode of a prefix of the same stream on one host core."""
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
    return {
        "value": round(sample_bytes / GIB / (t_enc + t_dec), 5),
        "unit": "GiB/s",
        "cores": 1,
        "kind": "port",
        "sample": "first %d MiB of the same splitmix64 stream, encode %.2f s + decode %.2f s, C oracle -O2 "
                  "(scalar per-symbol callbacks like reference source/huffman.c)" % (sample_bytes >> 20, t_enc, t_dec),
        "encode_GiBps": round(sample_bytes / GIB / t_enc, 5),
        "decode_GiBps": round(sample_bytes / GIB / t_dec, 5),
    }


def main():
    args = parse_args()
    ranks = Ranks(args.gpus)
    import harness  # the HIP library is loaded before anything else can pull in another HIP runtime

    lib = harness.load_product(args.library)
    if lib.aws_huffman_amd_device_count() < 1:
        raise SystemExit("bench.py: no HIP device visible and the product has no CPU path")
    patterns, lens = harness.load_table()
    coder = lib.aws_huffman_amd_table_coder_new(patterns, lens)
    ndev = lib.aws_huffman_amd_device_count()
    eng = harness.Engine(lib, coder, device=ranks.local_rank % ndev)

    n = args.bytes
    seed = 5 + ranks.rank  # cfg5 = seed 5 + g
    worst = n * 10 // 8 + 64
    d_in, d_enc, d_back = eng.alloc(n), eng.alloc(worst), eng.alloc(n + 64)
    eng.fill_splitmix64(d_in, n, seed)
    enc_plan = eng.encode_plan([dict(in_offset=0, in_len=n, out_offset=0, out_capacity=worst)])

    # untimed first pass: learn the encoded length, build the decode plan, check the round trip
    eng.encode_launch(enc_plan, d_in, d_enc)
    (rc, err, consumed, e_len, _, _), = eng.encode_results(enc_plan, 1)
    assert rc == 0 and consumed == n, (rc, err, consumed)
    dec_plan = eng.decode_plan([dict(in_offset=0, in_len=e_len, out_offset=0, out_capacity=n)])
    eng.decode_launch(dec_plan, d_enc, d_back)
    (rc, err, symbols, _), = eng.decode_results(dec_plan, 1)
    assert rc == 0 and symbols == n, (rc, err, symbols)
    import hashlib

    def digest(ptr, size):
        h = hashlib.sha256()
        for off in range(0, size, 256 << 20):
            h.update(eng.download(ptr, min(256 << 20, size - off), offset=off).tobytes())
        return h.hexdigest()

    in_digest = digest(d_in, n)
    assert digest(d_back, n) == in_digest, "round trip is not bit-exact"
    enc_digest = digest(d_enc, e_len)
    if n == GIB and seed == 5:
        pinned = harness.load_json("survey_probe_records.json")["streams"]["G1G"]
        assert (e_len, enc_digest) == (pinned["encoded_len"], pinned["sha256_encoded"]), "encoded stream differs from the reference's"

    def step(events_e=None, events_d=None):
        eng.encode_launch(enc_plan, d_in, d_enc, events=events_e)
        eng.decode_launch(dec_plan, d_enc, d_back, events=events_d)

    for _ in range(args.warmup):
        step()
    eng.sync()

    ev_e = [eng.new_events(4) for _ in range(args.steps)]
    ev_d = [eng.new_events(4) for _ in range(args.steps)]
    ranks.barrier()
    eng.sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(ev_e[k], ev_d[k])
    eng.sync()
    ranks.barrier()
    t1 = time.perf_counter()
    wall = ranks.max(t1 - t0)

    # per-kernel durations from the events recorded inside the timed region
    names_e = ["enc_count", "enc_scan", "enc_pack"]
    names_d = ["dec_sync", "dec_scan", "dec_emit"]
    kernel_ms = {k: 0.0 for k in names_e + names_d}
    for k in range(args.steps):
        for i, name in enumerate(names_e):
            kernel_ms[name] += eng.elapsed_ms(ev_e[k][i], ev_e[k][i + 1]) / args.steps
        for i, name in enumerate(names_d):
            kernel_ms[name] += eng.elapsed_ms(ev_d[k][i], ev_d[k][i + 1]) / args.steps
    t_enc_ms = sum(kernel_ms[k] for k in names_e)
    t_dec_ms = sum(kernel_ms[k] for k in names_d)

    # algorithmic bytes per launch (DESIGN.md "Kernels"): every input byte read once, every output byte written once
    algo_bytes = {
        "enc_count": n, "enc_scan": 0, "enc_pack": n + e_len,
        "dec_sync": e_len, "dec_scan": 0, "dec_emit": e_len + n,
    }
    dominant = max(kernel_ms, key=lambda k: kernel_ms[k])
    achieved = algo_bytes[dominant] / max(kernel_ms[dominant] * 1e-3, 1e-12)
    traffic = None
    pmc_path = os.path.join(REPO, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        traffic = json.load(open(pmc_path)).get(dominant, {}).get("hbm_bytes_per_launch")

    ms_per_step = wall / args.steps * 1e3
    total_units = ranks.sum(float(n)) / GIB  # GiB of input symbols per step, all ranks
    per_rank = ranks.gather({"rank": ranks.rank, "device": ranks.local_rank % ndev, "seed": seed,
                             "encoded_bytes": e_len, "sha256_encoded": enc_digest,
                             "encode_ms": round(t_enc_ms, 4), "decode_ms": round(t_dec_ms, 4)})
    out = {
        "metric": "GiB/s input consumed, encode+decode, 1 GiB random bytes; % HBM roofline",
        "value": round(total_units / (wall / args.steps), 6),
        "unit": "GiB/s",
        "n_gpus": ranks.world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {
            "workload": "1 GiB splitmix64 bytes per GPU (seed 5+rank), test_huffman_static_table coder: "
                        "encode (BASELINE configs[1]) then decode back (configs[2]), HBM-resident",
            "stream_bytes": n,
            "encoded_bytes": e_len,
            "bit_exact": True,
            "sha256_encoded": enc_digest,
        },
        "encode_GiBps": round(n / GIB / max(t_enc_ms * 1e-3, 1e-12), 2),
        "decode_GiBps_encoded_in": round(e_len / GIB / max(t_dec_ms * 1e-3, 1e-12), 2),
        "decode_GiBps_symbols_out": round(n / GIB / max(t_dec_ms * 1e-3, 1e-12), 2),
        "encode_path_frac_of_hbm_peak": round((n + e_len) / max(t_enc_ms * 1e-3, 1e-12) / HBM_PEAK_BYTES_PER_S, 4),
        "encode_read_frac_of_hbm_peak": round(n / max(t_enc_ms * 1e-3, 1e-12) / HBM_PEAK_BYTES_PER_S, 4),
        "decode_path_frac_of_hbm_peak": round((n + e_len) / max(t_dec_ms * 1e-3, 1e-12) / HBM_PEAK_BYTES_PER_S, 4),
        "ranks": per_rank,
        "kernel_ms": {k: round(v, 4) for k, v in kernel_ms.items()},
        "roofline": {
            "kernel": dominant,
            "bound": "hbm",
            "achieved": round(achieved / 1e9, 2),
            "peak": HBM_PEAK_BYTES_PER_S / 1e9,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_BYTES_PER_S, 4),
            "traffic": traffic,
            "algorithmic_bytes_per_launch": algo_bytes[dominant],
            "avg_launch_ms": round(kernel_ms[dominant], 4),
        },
    }
    if ranks.rank == 0 and ranks.world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(min(args.cpu_sample_mib << 20, n), seed)
    else:
        out["cpu_baseline"] = None
    if ranks.rank == 0:
        print(json.dumps(out), flush=True)
    ranks.close()


if __name__ == "__main__":
    main()
