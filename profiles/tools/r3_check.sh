# GPU box, round 3: the decode-road tests, then the stream bench line on every decode road
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "${1:-decode_roads or one_gib or one_shot or cut_streams}" > gpurun_out/r3_tests.log 2>&1; tail -15 gpurun_out/r3_tests.log
for mode in default one-pass; do
  AWS_HUFFMAN_AMD_DECODE=$mode timeout 300 python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/bench_$mode.json 2> gpurun_out/bench_$mode.err; tail -3 gpurun_out/bench_$mode.err
  python -c "
import json; d=json.load(open('gpurun_out/bench_$mode.json')); print('stream $mode', d['value'], d['kernel_ms'])"
done
