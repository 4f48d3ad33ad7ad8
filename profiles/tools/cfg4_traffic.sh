#!/bin/bash
# GPU box: HBM traffic of the cfg4 workload's kernels (BASELINE configs[3]) -- FETCH_SIZE and WRITE_SIZE in separate passes, as
# for the stream (profile_round.sh; MI355X_MICROARCH.md "HBM").  The per-dispatch tables are large: summarised here, on the
# box, into gpurun_out/<tag>/pmc_traffic_cfg4.json (copied to profiles/ by hand: bench.py reads it for the cfg4 leg).
#   usage: bash profiles/tools/cfg4_traffic.sh <tag> [buffer_bytes]
set -u
TAG=${1:-cfg4_traffic}
SIZE=${2:-16384}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --workload cfg4 --buffer-bytes $SIZE --steps 3 --warmup 1 --no-cpu-baseline --no-fresh"
cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o run -- $BENCH > /dev/null 2> "$OUT/pmc_fetch.err"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o run -- $BENCH > /dev/null 2> "$OUT/pmc_write.err"
python3 - "$ROOT" "$OUT" "$SIZE" <<'PY'
import json, os, sys
root, out, size = sys.argv[1], sys.argv[2], sys.argv[3]
sys.path.insert(0, os.path.join(root, "profiles", "tools"))
import summarise_profile as sp
counters = {}
for sub in ("pmc_fetch", "pmc_write"):
    sums, calls = sp.counter_sums(os.path.join(out, sub))
    steps = max([v for k, v in calls.items() if "dec_scan_small_kernel" in k] or [0])  # one per decode launch
    for g, cs in sums.items():
        for cname, total in cs.items():
            counters.setdefault(g, {})[cname] = total / max(steps, 1)
traffic = {"_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --workload cfg4 --buffer-bytes %s (profiles/tools/cfg4_traffic.sh), "
                      "per step = encode launch + resume launch + decode launch; FETCH doubled per the gfx950 correction" % size,
           "_steps_averaged": steps}
for g, cs in counters.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        traffic[g] = {"fetch_size_raw_kb": round(cs["FETCH_SIZE"], 1), "write_size_raw_kb": round(cs["WRITE_SIZE"], 1),
                      "hbm_read_bytes_per_launch": int(cs["FETCH_SIZE"] * 2048), "hbm_write_bytes_per_launch": int(cs["WRITE_SIZE"] * 1024),
                      "hbm_bytes_per_launch": int(cs["FETCH_SIZE"] * 2048 + cs["WRITE_SIZE"] * 1024)}
json.dump(traffic, open(os.path.join(out, "pmc_traffic_cfg4.json" if size == "16384" else "pmc_traffic_cfg4_%s.json" % size), "w"), indent=1, sort_keys=True)
print(json.dumps(traffic, indent=1))
PY
rm -rf "$OUT/pmc_fetch" "$OUT/pmc_write"
