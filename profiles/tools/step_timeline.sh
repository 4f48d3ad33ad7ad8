#!/bin/bash
# GPU box: every kernel (and fill / copy command) of the LAST timed step of a bench.py workload, in the order they ran, with the
# gaps between them and the queue each ran on -- what a step costs besides its big kernels.
#   usage: bash profiles/tools/step_timeline.sh <tag> [bench.py arguments, e.g. --workload cfg4 --no-fresh]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/step_timeline_$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o run -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --stage-events-every 1000 "$@" > "$OUT/bench.json" 2> "$OUT/err.txt"
python3 - "$OUT" <<'PY'
import csv, glob, re, sys
rows = []
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"(\w+_kernel(<[^>]*>)?)", n)
    return m.group(1) if m else n[:60]
names = [short(r["Kernel_Name"]) for r in rows]
# a step starts with the first enc_* kernel behind a dec_* kernel; the last whole step is the one before the last such start
starts = [i for i, n in enumerate(names) if n.startswith("enc_") and i and not names[i - 1].startswith("enc_")]
if len(starts) < 2:
    starts = [0, len(rows)]
lo, hi = starts[-2], starts[-1]
t0 = int(rows[lo]["Start_Timestamp"]); prev_end = t0
out = open(sys.argv[1] + "/timeline.txt", "w")
small = 0.0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    line = "%9.1f us  +%6.1f gap  %8.1f us  q%-3s %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), short(r["Kernel_Name"]))
    print(line); out.write(line + "\n")
    prev_end = max(prev_end, e)
tail = "step: %d commands, %.1f us from the first start to the last end" % (hi - lo, (prev_end - t0) / 1e3)
print(tail); out.write(tail + "\n")
PY
find "$OUT" -name '*kernel_trace.csv' -delete
