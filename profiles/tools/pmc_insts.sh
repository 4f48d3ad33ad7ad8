set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/insts
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs"
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/a -o run -- $BENCH > /dev/null 2> $OUT/a.err
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/b -o run -- $BENCH > /dev/null 2> $OUT/b.err
tail -3 $OUT/b.err
find $OUT -name '*kernel_trace.csv' -delete
