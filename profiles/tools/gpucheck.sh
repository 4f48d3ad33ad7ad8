# GPU box: the -m gpu suite, then the default bench line, and the same with the kernels the defaults replaced.
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; tail -3 gpurun_out/gpu_tests.log
timeout 300 python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -3 gpurun_out/bench.err
python -c "
import json; d=json.load(open('gpurun_out/bench.json')); print('default', d['value'], d['kernel_ms'])"
AWS_HUFFMAN_AMD_ENCODE=three-kernel AWS_HUFFMAN_AMD_DECODE=old-sync timeout 300 python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/bench3.json 2> gpurun_out/bench3.err
python -c "
import json; d=json.load(open('gpurun_out/bench3.json')); print('three-kernel encode, old sync', d['value'], d['kernel_ms'])"
