#!/bin/bash
# GPU box: the coder survey under rocprofv3 --kernel-trace: when the survey (host clock around one launch + wait)
# flags a launch as much slower than the median, do the kernels' own durations show it too?
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/survey_outliers
mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 $ROOT/profiles/tools/coder_survey.py > "$OUT/survey.txt" 2> "$OUT/stats.err"
grep -c . "$OUT/survey.txt"; grep "slowest" "$OUT/survey.txt"
python3 - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        calls, avg, mx = int(r["Calls"]), float(r["AverageNs"]), float(r["MaxNs"])
        if calls >= 7 and avg > 50000:
            print("%-60s calls %4d avg %9.1f us max %9.1f us  max/avg %.2f" % (r["Name"][:60], calls, avg / 1e3, mx / 1e3, mx / avg))
PY
find "$OUT" -name '*kernel_trace.csv' -size +4M -delete
