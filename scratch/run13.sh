#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for v in "$@"; do
  timeout -k 10 300 python -u scratch/bench_variant.py scratch/lib/libvsf_$v.so --no-cpu > gpurun_out/r13_bench_$v.json 2> gpurun_out/r13_bench_$v.err || { tail -5 gpurun_out/r13_bench_$v.err; exit 1; }
  python -c "import json;d=json.load(open('gpurun_out/r13_bench_$v.json'));print('$v', round(d['value']), {k:round(x,3) for k,x in d['kernel_ms'].items()})"
done
