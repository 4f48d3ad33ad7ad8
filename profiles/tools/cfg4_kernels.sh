#!/bin/bash
# GPU box: kernel durations of the cfg4 workload (65 536 buffers of 16 KiB): which kernels the batch's time is in
set -u
TAG=${1:-cfg4_kernels}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 $ROOT/bench.py --workload cfg4 --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/stats.err"
python3 - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 0.3:
            print("%-70s calls %4s avg %9.1f us  %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
find "$OUT" -name '*kernel_trace.csv' -size +4M -delete
