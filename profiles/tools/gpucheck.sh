mkdir -p gpurun_out && timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; tail -2 gpurun_out/gpu_tests.log; timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench.json 2> gpurun_out/bench.err; python -c "
import json; d=json.load(open('gpurun_out/bench.json')); print(d['value'], d['kernel_ms'])"
