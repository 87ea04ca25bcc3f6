#!/bin/bash
# C4 + ANMS and C2 + ANMS bench lines of the named library variants
set -o pipefail
mkdir -p gpurun_out
for v in "$@"; do
  timeout -k 10 300 python -u scratch/bench_variant.py scratch/lib/libvsf_$v.so --no-cpu --config C4 --detector anms > gpurun_out/r15_c4anms_$v.json 2> gpurun_out/r15_c4anms_$v.err || { tail -5 gpurun_out/r15_c4anms_$v.err; exit 1; }
  timeout -k 10 300 python -u scratch/bench_variant.py scratch/lib/libvsf_$v.so --no-cpu --detector anms > gpurun_out/r15_c2anms_$v.json 2> gpurun_out/r15_c2anms_$v.err || { tail -5 gpurun_out/r15_c2anms_$v.err; exit 1; }
  python -c "
import json
for f in ('c4anms','c2anms'):
    e=json.load(open('gpurun_out/r15_%s_$v.json'%f))
    print('$v', f, round(e['value']), {k:round(x,3) for k,x in e['kernel_ms'].items()})"
done
