#!/bin/bash
# GPU box: the kernels of ONE decode launch of the 1 GiB stream in the order they ran, with the gaps between them
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/decode_timeline
mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o run -- python3 $ROOT/profiles/tools/dec_only.py 4 > "$OUT/run.txt" 2> "$OUT/err.txt"
python3 - "$OUT" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the last decode launch: from the last dec_sync main kernel on
import re
def short(n):
    m = re.search(r"(\w+_kernel(<[^>]*>)?)", n)
    return m.group(1) if m else n[:60]
last = max(i for i, n in enumerate(names) if short(n).startswith("dec_sync_one_mixed_kernel") or (short(n).startswith("dec_sync_one_kernel") and "false" in short(n)))
t0 = int(rows[last]["Start_Timestamp"]); prev_end = t0
out = open(sys.argv[1] + "/timeline.txt", "w")
for r in rows[last:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    line = "%9.1f us  +%6.1f gap  %8.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, short(r["Kernel_Name"]))
    print(line); out.write(line + "\n")
    prev_end = e
PY
find "$OUT" -name '*kernel_trace.csv' -delete
