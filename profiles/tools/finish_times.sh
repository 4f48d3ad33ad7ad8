set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/finish_times
mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o run -- python3 $ROOT/bench.py --workload cfg4 --steps 2 --warmup 1 --no-cpu-baseline --no-fresh > "$OUT/bench.json" 2> "$OUT/err.txt"
python3 - "$OUT" <<'PY'
import csv, glob, re, sys
rows = []
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    n = r["Kernel_Name"]
    if "enc_finish" in n or "enc_onepass" in n or "enc_tiny" in n:
        print("%8.1f us grid %8s  %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")), re.search(r"(\w+_kernel(<[^>]*>)?)", n).group(1)))
PY
find "$OUT" -name '*kernel_trace.csv' -delete
