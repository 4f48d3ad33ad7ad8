# GPU box, round 4: the stream with shares of dec_emit_fast's workgroups taking their table entries through memory
mkdir -p gpurun_out
for share in 0/1 1/8 1/6 1/5 1/4 1/3; do
  name=$(echo $share | tr / _)
  AWS_HUFFMAN_AMD_EMIT_VIA_MEMORY=$share timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps 8 > gpurun_out/bench_via_$name.json 2> gpurun_out/bench_via_$name.err; tail -2 gpurun_out/bench_via_$name.err
  python -c "
import json; d=json.load(open('gpurun_out/bench_via_$name.json')); print('emit via memory $share', d['value'], d['kernel_ms'], d['config'].get('bit_exact'))"
done
