#!/bin/bash
# LSSD bench line of the named library variants
set -o pipefail
mkdir -p gpurun_out
for v in "$@"; do
  timeout -k 10 300 python -u scratch/bench_variant.py scratch/lib/libvsf_$v.so --no-cpu --tracker lssd > gpurun_out/r18_lssd_$v.json 2> gpurun_out/r18_lssd_$v.err || { tail -5 gpurun_out/r18_lssd_$v.err; exit 1; }
  python -c "
import json
e=json.load(open('gpurun_out/r18_lssd_$v.json'))
print('$v', 'LSSD', round(e['value']), {k:round(x,3) for k,x in e['kernel_ms'].items()})"
done
