#!/usr/bin/env python3
"""Turns gpurun_out/<tag>/ (written by profile_round.sh on the GPU box) into the files kept
under profiles/:

  profiles/<name>_bench.json          the default bench line of that run
  profiles/<name>_kernel_stats.csv    rocprofv3 --kernel-trace --stats summary of bench.py
  profiles/<name>_counters.json       per bench-kernel PMC sums per launch
  profiles/pmc_traffic.json           HBM bytes per launch per bench kernel (bench.py reads it
                                      for roofline.traffic)

HBM bytes follow MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in KiB-like units of
1024 B, collected in separate passes; on gfx950 FETCH_SIZE reports half the bytes of a wide
coalesced streaming read, so the read side is doubled (all our big reads are 16 B per lane
global loads or LDS-DMA); WRITE_SIZE is exact for 16 B per lane streaming stores.

  usage: summarise_profile.py <tag> <name> [traffic-file]     e.g.  r01d r01_d_lookback
  (traffic-file: where the HBM bytes go instead of profiles/pmc_traffic.json, for a profile of a road that is not the default)
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

GROUPS = [  # bench kernel name <- substrings of the device kernel names it covers
    ("enc_onepass", ["enc_onepass_kernel"]),
    ("enc_finish", ["enc_finish_kernel"]),
    ("enc_tiny", ["enc_tiny_kernel"]),
    ("enc_count", ["enc_count_kernel"]),
    ("enc_scan", ["enc_scan_small_kernel", "enc_scan_large_kernel"]),
    ("enc_pack", ["enc_pack_wave_kernel", "enc_pack_stream_kernel", "enc_pack_kernel"]),
    ("dec_sync", ["dec_sync_one_kernel", "dec_sync_one_mixed_kernel", "dec_sync_pack_kernel", "dec_sync_guess_kernel", "dec_sync_few_kernel", "dec_sync_tail_kernel",
                  "dec_sync_kernel"]),
    ("dec_scan", ["dec_scan_small_kernel", "dec_scan_runs_kernel", "dec_scan_apply_kernel",
                  "dec_sync_true_kernel", "dec_tiny_kernel", "dec_deep_kernel"]),
    ("dec_emit", ["dec_emit_fast_kernel", "dec_emit_pack_kernel", "dec_emit_tail_kernel", "dec_emit_big_kernel", "dec_emit_kernel"]),
]


ONE_PASS = False  # set by counter_sums: the profile holds enc_onepass_kernel dispatches


def group_of(kernel_name):
    for g, subs in GROUPS:
        if any(s in kernel_name for s in subs):
            return g
    return None


def counter_sums(directory):
    """{group: {counter: sum over dispatches}}, {group: dispatches of its first kernel}"""
    sums = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    global ONE_PASS
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    ONE_PASS = any("enc_onepass_kernel" in r["Kernel_Name"] for f in files for r in csv.DictReader(open(f)))
    for f in files:
        seen = set()
        for r in csv.DictReader(open(f)):
            g = group_of(r["Kernel_Name"])
            if g is None:
                continue
            sums[g][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (r["Dispatch_Id"], r["Kernel_Name"])
            if key not in seen:
                seen.add(key)
                calls[r["Kernel_Name"]] += 1
    return sums, calls


def main():
    tag, name = sys.argv[1], sys.argv[2]
    src = os.path.join(REPO, "gpurun_out", tag)
    dst = os.path.join(REPO, "profiles")
    shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, name + "_bench.json"))
    stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(dst, name + "_kernel_stats.csv"))
    counters = {}
    steps = None
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
        sums, calls = counter_sums(os.path.join(src, sub))
        # one per decode launch (the 1 GiB stream: dec_scan_runs; a plan without long items: dec_scan_small)
        n_steps = max([v for k, v in calls.items() if "dec_scan_runs_kernel" in k or "dec_scan_small_kernel" in k] or [0])
        if not n_steps:
            continue
        steps = n_steps
        for g, cs in sums.items():
            for cname, total in cs.items():
                counters.setdefault(g, {})[cname] = total / n_steps
    traffic = {}
    for g, cs in counters.items():
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            read_b = cs["FETCH_SIZE"] * 1024 * 2  # gfx950: wide streaming reads are tallied at half
            write_b = cs["WRITE_SIZE"] * 1024
            traffic[g] = {
                "fetch_size_raw_kb": round(cs["FETCH_SIZE"], 1),
                "write_size_raw_kb": round(cs["WRITE_SIZE"], 1),
                "hbm_read_bytes_per_launch": int(read_b),
                "hbm_write_bytes_per_launch": int(write_b),
                "hbm_bytes_per_launch": int(read_b + write_b),
            }
        if "TCC_HIT_sum" in cs:
            cs["l2_hit_rate"] = cs["TCC_HIT_sum"] / max(cs["TCC_HIT_sum"] + cs["TCC_MISS_sum"], 1.0)
    json.dump({"profile": name, "launches_averaged": steps, "per_launch": counters},
              open(os.path.join(dst, name + "_counters.json"), "w"), indent=1, sort_keys=True)
    if traffic:
        traffic["_source"] = name + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py, FETCH doubled per the gfx950 correction)"
        json.dump(traffic, open(os.path.join(dst, sys.argv[3] if len(sys.argv) > 3 else "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
