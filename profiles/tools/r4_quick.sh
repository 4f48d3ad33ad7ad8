# GPU box, round 4: a parity subset first (fail fast), then the stream bench line with the default kernels and with the
# sync kernel the default replaced, then config 4.
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "${1:-roundtrips or garbage or cut_streams or damaged or config4 or second_sync or null_empty or one_gib}" > gpurun_out/gpu_subset.log 2>&1; tail -4 gpurun_out/gpu_subset.log
for mode in default lean-sync; do
  AWS_HUFFMAN_AMD_DECODE=$mode timeout 300 python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/bench_$mode.json 2> gpurun_out/bench_$mode.err; tail -2 gpurun_out/bench_$mode.err
  python -c "
import json; d=json.load(open('gpurun_out/bench_$mode.json')); print('stream $mode', d['value'], d['kernel_ms'], d['config'].get('bit_exact'))"
done
timeout 300 python bench.py --no-cpu-baseline --workload cfg4 --steps 5 --warmup 2 > gpurun_out/bench_cfg4.json 2> gpurun_out/bench_cfg4.err; tail -2 gpurun_out/bench_cfg4.err
python -c "
import json; d=json.load(open('gpurun_out/bench_cfg4.json')); print('cfg4', d['value'], d['kernel_ms'], d['config'].get('bit_exact'))"
