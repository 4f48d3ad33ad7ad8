#!/bin/bash
# GPU box: kernel durations behind profiles/tools/long_code_stream.py (one long stream of a coder with long codes)
set -u
NAME=${1:-hpack_lengths}; N=${2:-134217728}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/long_code_kernels
mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 $ROOT/profiles/tools/long_code_stream.py $NAME $N > "$OUT/out.txt" 2> "$OUT/stats.err"
cat "$OUT/out.txt"
python3 - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["AverageNs"]) > 20000:
            print("%-70s calls %4s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
find "$OUT" -name '*kernel_trace.csv' -size +4M -delete
