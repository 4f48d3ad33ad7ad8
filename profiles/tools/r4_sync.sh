# GPU box, round 4: the stream (and config 4) with the variants of the sync kernel
mkdir -p gpurun_out
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --steps 8 > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err; tail -2 gpurun_out/bench_$name.err
  python -c "
import json; d=json.load(open('gpurun_out/bench_$name.json')); print('stream $name', d['value'], d['kernel_ms'], d['config'].get('bit_exact'))"
}
run bank4 AWS_HUFFMAN_AMD_BANK_CHUNKS=4
run bank2 AWS_HUFFMAN_AMD_BANK_CHUNKS=2
run lean AWS_HUFFMAN_AMD_DECODE=lean-sync
for ch in 4 2; do
AWS_HUFFMAN_AMD_BANK_CHUNKS=$ch timeout 300 python bench.py --no-cpu-baseline --workload cfg4 --steps 5 --warmup 2 > gpurun_out/bench_cfg4_$ch.json 2> gpurun_out/bench_cfg4_$ch.err; tail -2 gpurun_out/bench_cfg4_$ch.err
python -c "
import json; d=json.load(open('gpurun_out/bench_cfg4_$ch.json')); print('cfg4 bank$ch', d['value'], d['kernel_ms'], d['config'].get('bit_exact'))"
done
