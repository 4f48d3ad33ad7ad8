# GPU box, round 4: dec_sync_resident against dec_sync_lean on the stream; parity with it forced onto small streams
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "resident or one_gib or roundtrips or damaged or garbage" > gpurun_out/gpu_subset.log 2>&1; tail -4 gpurun_out/gpu_subset.log
for mode in default lean-sync; do
  AWS_HUFFMAN_AMD_DECODE=$mode timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps 8 > gpurun_out/bench_$mode.json 2> gpurun_out/bench_$mode.err; tail -2 gpurun_out/bench_$mode.err
  python -c "
import json; d=json.load(open('gpurun_out/bench_$mode.json')); print('stream $mode', d['value'], d['kernel_ms'], d['config'].get('bit_exact'))"
done
