#!/bin/bash
# ANMS kernels with large workgroups: suite, then the two ANMS bench lines
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r22_tests.txt 2>&1 || { tail -30 gpurun_out/r22_tests.txt; exit 1; }
tail -2 gpurun_out/r22_tests.txt
timeout -k 10 600 python3 -u bench.py --no-cpu --config C4 --detector anms > gpurun_out/r22_C4_anms.json 2> gpurun_out/r22_C4_anms.err || { tail -5 gpurun_out/r22_C4_anms.err; exit 1; }
timeout -k 10 300 python3 -u bench.py --no-cpu --detector anms --tracker lssd > gpurun_out/r22_anms_lssd.json 2> gpurun_out/r22_anms_lssd.err || { tail -5 gpurun_out/r22_anms_lssd.err; exit 1; }
for f in C4_anms anms_lssd; do python3 -c "
import json
d=json.load(open('gpurun_out/r22_$f.json'))
print('$f', round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernel_ms'].items()}, 'feat', d['config']['mean_features_per_stream'], 'ver', round(d['config']['verified_fraction'],5))
"; done
