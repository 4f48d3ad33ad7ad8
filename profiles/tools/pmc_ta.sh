set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ta
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs"
timeout 600 rocprofv3 --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/a -o run -- $BENCH > /dev/null 2> $OUT/a.err
timeout 600 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TAGRAM0_REQ_sum TCP_TCC_READ_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum --kernel-trace --output-format csv -d $OUT/b -o run -- $BENCH > /dev/null 2> $OUT/b.err
tail -2 $OUT/b.err
find $OUT -name '*kernel_trace.csv' -delete
