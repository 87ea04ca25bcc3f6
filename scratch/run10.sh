#!/bin/bash
# detector parity tests with the in-tree library, then C2 and ANMS bench lines of the named library variants
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r10_tests.txt 2>&1 || { tail -30 gpurun_out/r10_tests.txt; exit 1; }
tail -2 gpurun_out/r10_tests.txt
for v in "$@"; do
  timeout -k 10 300 python -u scratch/bench_variant.py scratch/lib/libvsf_$v.so --no-cpu > gpurun_out/r10_bench_$v.json 2> gpurun_out/r10_bench_$v.err || { tail -5 gpurun_out/r10_bench_$v.err; exit 1; }
  timeout -k 10 300 python -u scratch/bench_variant.py scratch/lib/libvsf_$v.so --no-cpu --detector anms --tracker lssd > gpurun_out/r10_bench_anms_$v.json 2> gpurun_out/r10_bench_anms_$v.err || { tail -5 gpurun_out/r10_bench_anms_$v.err; exit 1; }
  python -c "
import json
d=json.load(open('gpurun_out/r10_bench_$v.json')); e=json.load(open('gpurun_out/r10_bench_anms_$v.json'))
print('$v', 'C2', round(d['value']), round(d['kernel_ms']['lk'],3), round(d['kernel_ms']['score'],3), 'ANMS', round(e['value']), round(e['kernel_ms']['score'],3))"
done
