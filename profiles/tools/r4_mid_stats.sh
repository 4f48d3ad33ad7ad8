# GPU box: per-kernel times of the mid-size batch (65 536 x 2 KiB in configs[3]'s shape)
mkdir -p gpurun_out/mid_stats
export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/mid_stats -o run -- python3 $ROOT/bench.py --workload cfg4 --buffer-bytes 2048 --steps 5 --warmup 2 --no-cpu-baseline > $ROOT/gpurun_out/mid_stats/bench.json 2> $ROOT/gpurun_out/mid_stats/bench.err
cd $ROOT
find gpurun_out/mid_stats -name '*kernel_trace.csv' -size +4M -delete
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/mid_stats/**/run_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:22]:
    print("%-70s calls %5s avg %9.1f us total %8.2f ms"%(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
