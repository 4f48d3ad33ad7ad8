#!/bin/bash
# GPU box: what the one-pass encoder's look-back costs.  The same 1 GiB as 262 144 items of one tile (enc_onepass<.., SOLO>:
# a wave an item, nobody waits for anybody) and as 65 536 items of one segment (enc_onepass: look-back over 262 144 tiles);
# kernel durations from rocprofv3.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/solo_vs_lookback
mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
MID_ITEMS_TOTAL_MIB=1024 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 $ROOT/profiles/tools/mid_items.py 4096 16384 > "$OUT/mid_items.txt" 2> "$OUT/stats.err"
cat "$OUT/mid_items.txt"
python3 - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "enc_" in r["Name"]:
            print("%-60s calls %4s avg %9.1f us min %9.1f max %9.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
find "$OUT" -name '*kernel_trace.csv' -size +4M -delete
