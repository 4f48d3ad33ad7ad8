# GPU box: the -m gpu suite, the stream and cfg4 bench lines, and the mid-size item batches
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; tail -3 gpurun_out/gpu_tests.log
timeout 300 python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -3 gpurun_out/bench.err
python -c "
import json; d=json.load(open('gpurun_out/bench.json')); print('stream', d['value'], d['kernel_ms'])"
timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --workload cfg4 > gpurun_out/bench_cfg4.json 2> gpurun_out/bench_cfg4.err; tail -3 gpurun_out/bench_cfg4.err
python -c "
import json; d=json.load(open('gpurun_out/bench_cfg4.json')); print('cfg4', d['value'], d['kernel_ms'])"
timeout 300 python profiles/tools/mid_items.py 2>&1 | tail -8
