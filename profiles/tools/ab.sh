#!/bin/bash
# GPU box: builds of the library side by side on ONE box (boxes differ by a few percent: only numbers of one call compare).
#   usage: bash profiles/tools/ab.sh <rounds> <steps> variant.so [variant.so ...]   (paths under the repository, e.g. ab/base.so)
# Every round runs the default bench line (no CPU baseline, no extra legs) once per variant (bench.py --library), in turn.
# Prints value, ms_per_step and the stage medians of every run, then each variant's best and median.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
ROUNDS=$1; STEPS=$2; shift 2
OUT=gpurun_out/ab; mkdir -p "$OUT"
EXTRA=${AB_BENCH_ARGS:-}
for r in $(seq 1 "$ROUNDS"); do
    for v in "$@"; do
        name=$(basename "$v" .so)
        timeout 300 python3 bench.py --library "$ROOT/$v" --no-cpu-baseline --no-extra-legs --steps "$STEPS" --warmup 5 $EXTRA > "$OUT/${name}_$r.json" 2> "$OUT/${name}_$r.err" || tail -3 "$OUT/${name}_$r.err"
    done
done
python3 - "$OUT" "$ROUNDS" "$@" <<'PY'
import json, os, statistics, sys
out, rounds = sys.argv[1], int(sys.argv[2])
for v in sys.argv[3:]:
    name = os.path.basename(v)[:-3]
    vals, steps = [], []
    for r in range(1, rounds + 1):
        try:
            d = json.loads(open("%s/%s_%d.json" % (out, name, r)).read().strip().splitlines()[-1])
        except Exception as e:
            print(name, r, "no line:", e); continue
        vals.append(d["value"]); steps.append(d["ms_per_step"])
        print("%-22s round %d  value %7.1f  ms_per_step %.4f  %s  bit_exact %s" % (
            name, r, d["value"], d["ms_per_step"], {k: round(x, 4) for k, x in d.get("kernel_ms", {}).items()}, d.get("config", {}).get("bit_exact")))
    if vals:
        print("%-22s best %.1f  median %.1f GiB/s   ms_per_step best %.4f median %.4f" % (
            name, max(vals), statistics.median(vals), min(steps), statistics.median(steps)))
PY
