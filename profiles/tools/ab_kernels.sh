#!/bin/bash
# GPU box: kernel durations (rocprofv3 --kernel-trace --stats) of the default bench line for builds of the library side by side.
#   usage: bash profiles/tools/ab_kernels.sh variant.so [variant.so ...]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
for v in "$@"; do
    name=$(basename "$v" .so)
    OUT=$ROOT/gpurun_out/ab_kernels/$name; mkdir -p "$OUT"
    cd /tmp
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o run -- python3 $ROOT/bench.py --library "$ROOT/$v" --steps 12 --warmup 3 --no-cpu-baseline --no-extra-legs > "$OUT/bench.json" 2> "$OUT/err.txt"
    cd "$ROOT"
    find "$OUT" -name '*kernel_trace.csv' -delete
    python3 - "$OUT" "$name" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/run_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print("%-18s %-64s calls %3s avg %8.1f us min %8.1f" % (sys.argv[2], r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
