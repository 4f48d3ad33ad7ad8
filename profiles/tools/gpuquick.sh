# GPU box: stream bench line with the default kernels (optionally a decode mode in $1)
mkdir -p gpurun_out
for mode in default ${1:-}; do
  AWS_HUFFMAN_AMD_DECODE=$mode timeout 300 python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/bench_$mode.json 2> gpurun_out/bench_$mode.err; tail -3 gpurun_out/bench_$mode.err
  python -c "
import json; d=json.load(open('gpurun_out/bench_$mode.json')); print('stream $mode', d['value'], d['kernel_ms'])"
done
