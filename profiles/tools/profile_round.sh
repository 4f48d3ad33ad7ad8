#!/bin/bash
# Runs on the GPU box (gpurun): GPU parity tests, the default bench line, a rocprofv3
# kernel-trace/stats pass and two PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass:
# MI355X_MICROARCH.md "rocprofv3 PMC slots") of the same bench command.  Everything lands in
# gpurun_out/<tag>/; profiles/tools/summarise_profile.py turns it into the files committed
# under profiles/.
#   usage: bash profiles/tools/profile_round.sh <tag> [skip-tests]
set -u
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
if [ "${2:-}" != "skip-tests" ]; then
    timeout 1200 python3 -m pytest tests -m gpu -x -q > "$OUT/gpu_tests.log" 2>&1
    tail -2 "$OUT/gpu_tests.log"
fi
timeout 600 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 600 "$OUT/bench.json"
timeout 600 python3 bench.py --workload cfg4 --no-cpu-baseline > "$OUT/bench_cfg4.json" 2> "$OUT/bench_cfg4.err"
timeout 600 python3 bench.py --workload host-abi --no-cpu-baseline > "$OUT/bench_host_abi.json" 2> "$OUT/bench_host_abi.err"
tail -c 300 "$OUT/bench_cfg4.json"; tail -c 300 "$OUT/bench_host_abi.json"
BENCH="python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- $BENCH > "$OUT/stats.json" 2> "$OUT/stats.err"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o run -- $BENCH > /dev/null 2> "$OUT/pmc_fetch.err"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o run -- $BENCH > /dev/null 2> "$OUT/pmc_write.err"
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_sq" -o run -- $BENCH > /dev/null 2> "$OUT/pmc_sq.err"
# the raw per-dispatch traces are large: keep the stats and the counter tables
find "$OUT" -name '*kernel_trace.csv' -size +4M -delete
ls -R "$OUT" | head -40
