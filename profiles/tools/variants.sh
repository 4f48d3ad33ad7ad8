# GPU box: the bench line's kernel times for the default build and for every library under aws-c-compression_amd/variants/
mkdir -p gpurun_out
for lib in default aws-c-compression_amd/variants/*.so; do
    if [ "$lib" = default ]; then arg=""; else arg="--library $lib"; fi
    timeout 300 python bench.py --no-cpu-baseline --no-extra-legs $arg > gpurun_out/bench_v.json 2> gpurun_out/bench_v.err || tail -3 gpurun_out/bench_v.err
    python -c "
import json,sys; d=json.load(open('gpurun_out/bench_v.json')); print('$lib', d['value'], d['kernel_ms'])"
done
