#!/bin/bash
# GPU box: SQ counters of the cfg4 workload's kernels (what its end-of-stream kernels do differently from the stream's)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/cfg4_counters
mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$OUT/pmc" -o run -- python3 $ROOT/bench.py --workload cfg4 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> "$OUT/pmc.err"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$OUT/pmc_stream" -o run -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs > /dev/null 2> "$OUT/pmc_stream.err"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
for tag in ("pmc", "pmc_stream"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for f in glob.glob(sys.argv[1] + "/" + tag + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "dec_sync_one" in k or "dec_sync_pack" in k or "dec_emit_fast" in k:
                name = ("sync_pack" if "sync_pack" in k else "sync_one" if "sync_one" in k else "emit") + ("<TAIL>" if ("ELb1E" in k or ", true" in k) else "")
                acc[name][r["Counter_Name"]] += float(r["Counter_Value"]); 
                if r["Counter_Name"] == "SQ_WAVES": calls[name] += 1
    for name in sorted(acc):
        n = max(calls[name], 1)
        print(tag, name, "launches", n, {k: round(v / n / 1e6, 2) for k, v in sorted(acc[name].items())})
PY
# (the per-dispatch tables of a workload that fills 65 536 buffers one launch each are tens of MB: the summary above is what is kept)
rm -rf "$OUT/pmc" "$OUT/pmc_stream"
