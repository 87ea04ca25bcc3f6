#!/bin/bash
# experiment build against the committed one: C2 and ANMS score times
set -o pipefail
mkdir -p gpurun_out
for v in exp base; do
  lib=scratch/lib/libvsf_hip_exp.so; [ $v = base ] && lib=vi-slam_amd/lib/libvsf_hip.so
  timeout -k 10 200 python3 -u scratch/bench_variant.py $lib --no-cpu > gpurun_out/r20_${v}_C2.json 2> gpurun_out/r20_${v}_C2.err || { tail -5 gpurun_out/r20_${v}_C2.err; exit 1; }
  timeout -k 10 200 python3 -u scratch/bench_variant.py $lib --no-cpu --detector anms --tracker lssd > gpurun_out/r20_${v}_anms.json 2> gpurun_out/r20_${v}_anms.err || { tail -5 gpurun_out/r20_${v}_anms.err; exit 1; }
  for f in C2 anms; do python3 -c "
import json
d=json.load(open('gpurun_out/r20_${v}_$f.json'))
print('$v $f', round(d['value']), round(d['ms_per_step'],3), 'score', round(d['kernel_ms']['score'],4), 'lk', round(d['kernel_ms']['lk'],3), 'feat', d['config']['mean_features_per_stream'], 'ver', round(d['config']['verified_fraction'],5))
"; done
done
