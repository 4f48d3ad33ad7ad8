#!/bin/bash
# GPU box: the stream bench with the one-pass decoder (AWS_HUFFMAN_AMD_DECODE=one-pass): bench line, kernel stats, HBM traffic
set -u
TAG=${1:-onepass}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp; cd "$ROOT"
export AWS_HUFFMAN_AMD_DECODE=one-pass
timeout 600 python3 bench.py --no-cpu-baseline --no-extra-legs > "$OUT/bench.json" 2> "$OUT/bench.err"; tail -c 400 "$OUT/bench.json"
BENCH="python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- $BENCH > /dev/null 2> "$OUT/stats.err"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o run -- $BENCH > /dev/null 2> "$OUT/pmc_fetch.err"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o run -- $BENCH > /dev/null 2> "$OUT/pmc_write.err"
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_sq" -o run -- $BENCH > /dev/null 2> "$OUT/pmc_sq.err"
find "$OUT" -name '*kernel_trace.csv' -size +4M -delete
