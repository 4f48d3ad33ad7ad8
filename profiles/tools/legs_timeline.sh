#!/bin/bash
# GPU box: the last commands of the default bench line's run -- its extra legs (cfg4, mid_items, header_items, host_abi) -- in the
# order they ran, with gaps and queues: what those legs' steps are made of.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/legs_timeline
mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o run -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/err.txt"
python3 - "$OUT" <<'PY'
import csv, glob, re, sys
rows = []
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"(\w+_kernel(<[^>]*>)?)", n)
    return m.group(1) if m else n[:40]
out = open(sys.argv[1] + "/timeline.txt", "w")
# the last commands of the run: the legs come behind the stream's timed steps, one after the other (cfg4, mid_items, header_items, host_abi)
last_stream = max(0, len(rows) - int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) - 160)
prev_end = int(rows[last_stream]["Start_Timestamp"]); t0 = prev_end
for r in rows[last_stream:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s - prev_end > 2_000_000:
        out.write("   ---- %.1f ms later\n" % ((s - prev_end) / 1e6))
    out.write("%10.1f us +%7.1f gap %8.1f us q%-3s %s\n" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), short(r["Kernel_Name"])))
    prev_end = max(prev_end, e)
PY
find "$OUT" -name '*kernel_trace.csv' -delete
