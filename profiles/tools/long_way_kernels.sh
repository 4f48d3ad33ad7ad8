#!/bin/bash
# GPU box: kernel durations behind profiles/tools/other_distributions.py (the streams that take the long way)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/long_way_kernels
mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 $ROOT/profiles/tools/other_distributions.py > "$OUT/out.txt" 2> "$OUT/stats.err"
cat "$OUT/out.txt"
python3 - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 0.5 and "dec_" in r["Name"]:
            print("%-75s calls %4s avg %9.1f us min %9.1f max %9.1f" % (r["Name"][:75], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
find "$OUT" -name '*kernel_trace.csv' -size +4M -delete
