# GPU box: SQ counters of the decode kernels on the stream bench (three PMC passes), summed per kernel
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3_counters
rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs"
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/a -o run -- $BENCH > /dev/null 2> $OUT/a.err
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/b -o run -- $BENCH > /dev/null 2> $OUT/b.err
timeout 600 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $OUT/c -o run -- $BENCH > /dev/null 2> $OUT/c.err
tail -2 $OUT/c.err
find $OUT -name '*kernel_trace.csv' -delete
cd $ROOT
python3 - <<'PY'
import csv, glob, collections, json, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r3_counters")
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        k = k.replace("(anonymous namespace)::", "").replace("void ", "")
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(k, row["Counter_Name"])] += 1
res = {}
for k, d in tot.items():
    if not (k.startswith("dec_") or k.startswith("enc_onepass")):
        continue
    res[k] = {c: v / max(cnt[(k, c)], 1) for c, v in d.items()}  # per dispatch
json.dump(res, open(out + "/summary.json", "w"), indent=1, sort_keys=True)
for k, d in sorted(res.items()):
    if d.get("SQ_WAVE_CYCLES", 0) < 1e6:
        continue
    print(k)
    print("   " + " ".join("%s=%.3g" % (c.replace("SQ_", ""), v) for c, v in sorted(d.items())))
PY
