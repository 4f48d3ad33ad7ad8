# GPU box: the default bench line only (no CPU baseline), one-pass and three-kernel encoders; optional quick parity subset first
mkdir -p gpurun_out
if [ "${1:-}" = "tests" ]; then
    timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; tail -3 gpurun_out/gpu_tests.log
fi
timeout 300 python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -3 gpurun_out/bench.err
python -c "
import json; d=json.load(open('gpurun_out/bench.json')); print('default', d['value'], d['kernel_ms'])"
