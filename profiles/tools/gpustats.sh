# GPU box: per-kernel times of the default bench line (rocprofv3 --kernel-trace --stats), top rows
mkdir -p gpurun_out/stats
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/stats -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra-legs --steps 5 --warmup 1 ${1:-} > $GRAFT_REPO_ROOT/gpurun_out/stats/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/stats/bench.err
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/stats -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-60s calls %6s total %10.3f ms avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
