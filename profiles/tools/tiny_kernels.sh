mkdir -p gpurun_out/tiny_stats; export TMPDIR=/tmp; ROOT=$(pwd); cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/tiny_stats -o run -- python3 $ROOT/profiles/tools/tiny_items.py > $ROOT/gpurun_out/tiny_stats/out.txt 2>&1
cd $ROOT; find gpurun_out/tiny_stats -name '*kernel_trace.csv' -size +4M -delete
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/tiny_stats/**/run_kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print("%-60s calls %4s avg %8.1f us min %8.1f"%(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
