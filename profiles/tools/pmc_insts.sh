#!/bin/bash
# GPU box: instruction and busy-cycle counters of the default bench line's kernels (two rocprofv3 --pmc passes), summarised per kernel and launch
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/insts
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs"
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/a -o run -- $BENCH > /dev/null 2> $OUT/a.err
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/b -o run -- $BENCH > /dev/null 2> $OUT/b.err
tail -3 $OUT/b.err
find $OUT -name '*kernel_trace.csv' -delete
# per kernel and launch, in millions (quad-cycles for the ACTIVE / CYCLES counters): what DESIGN.md's "vector units busy" figures come from
python3 - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import collections, csv, glob, sys
for sub in ("a", "b"):
    for f in glob.glob(sys.argv[1] + "/%s/**/*counter_collection.csv" % sub, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:44]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k].add(r["Dispatch_Id"])
        for k, v in acc.items():
            n = len(calls[k])
            if max(v.values()) / n > 2e6:
                print("%s %-44s launches %2d  " % (sub, k, n) + "  ".join("%s %.2f" % (c.replace("SQ_", ""), x / n / 1e6) for c, x in sorted(v.items())))
PY
