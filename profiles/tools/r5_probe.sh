# GPU box, round 5: the small-table walk probe (profiles/tools/micro/probe_r05.hip), then the new plan scenario and a stream bench
mkdir -p gpurun_out/r05
timeout 300 profiles/tools/micro/build/probe_r05 > gpurun_out/r05/probe_walk_small_table.jsonl 2>&1; echo "probe rc $?"
cat gpurun_out/r05/probe_walk_small_table.jsonl
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "two_last_chunks or mid_sized" 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline --no-extra-legs > gpurun_out/r05/bench_start.json 2> gpurun_out/r05/bench_start.err; tail -2 gpurun_out/r05/bench_start.err
python -c "
import json; d=json.load(open('gpurun_out/r05/bench_start.json')); print('stream', d['value'], d['kernel_ms'])"
