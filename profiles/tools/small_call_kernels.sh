#!/bin/bash
# GPU box: the kernels behind profiles/tools/small_call_latency.py (how long the one launch of a small call runs)
set -u
TAG=${1:-small_calls}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 $ROOT/profiles/tools/small_call_latency.py > "$OUT/latency.txt" 2> "$OUT/stats.err"
cat "$OUT/latency.txt"
python3 - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-60s calls %6s avg %8.1f us min %8.1f max %8.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
find "$OUT" -name '*kernel_trace.csv' -size +4M -delete
