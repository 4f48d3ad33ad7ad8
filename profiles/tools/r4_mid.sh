# GPU box, round 4: the mid_items tool and the bench line's legs (cfg4, mid_items) with the default kernels and with a workgroup per end-of-stream chunk (lean-sync)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mid_sized or config4 or tiny or batched" > gpurun_out/gpu_subset.log 2>&1; tail -3 gpurun_out/gpu_subset.log
for mode in default lean-sync; do
  echo "mode $mode" >> gpurun_out/mid.txt
  AWS_HUFFMAN_AMD_DECODE=$mode timeout 600 python profiles/tools/mid_items.py 1024 2048 4096 8192 12288 >> gpurun_out/mid.txt 2>&1
  AWS_HUFFMAN_AMD_DECODE=$mode timeout 300 python bench.py --no-cpu-baseline --steps 5 > gpurun_out/bench_$mode.json 2> gpurun_out/bench_$mode.err; tail -2 gpurun_out/bench_$mode.err
  python -c "
import json; d=json.load(open('gpurun_out/bench_$mode.json')); print('$mode stream', d['value'], d['kernel_ms']); [print(' ', leg, d[leg]['value_GiBps'], d[leg]['kernel_ms']) for leg in ('cfg4','mid_items')]"
done
cat gpurun_out/mid.txt
