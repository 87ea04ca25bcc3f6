#!/bin/bash
# LK variant against the committed build: C2 line (twice each, alternating)
set -o pipefail
mkdir -p gpurun_out
for v in exp base exp base; do
  lib=scratch/lib/libvsf_hip_exp.so; [ $v = base ] && lib=vi-slam_amd/lib/libvsf_hip.so
  timeout -k 10 200 python3 -u scratch/bench_variant.py $lib --no-cpu > gpurun_out/r24_${v}.json 2> gpurun_out/r24_${v}.err || { tail -5 gpurun_out/r24_${v}.err; exit 1; }
  python3 -c "
import json
d=json.load(open('gpurun_out/r24_${v}.json'))
print('$v', round(d['value']), round(d['ms_per_step'],3), 'lk', round(d['kernel_ms']['lk'],4), 'score', round(d['kernel_ms']['score'],3), 'feat', d['config']['mean_features_per_stream'], 'ver', round(d['config']['verified_fraction'],6))
"
done
