#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 200 python -u scratch/lk_stats.py 256 0.06 > gpurun_out/r8_lkstats.json 2> gpurun_out/r8_lkstats.err || { tail -5 gpurun_out/r8_lkstats.err; exit 1; }
python -c "
import json
d=json.load(open('gpurun_out/r8_lkstats.json'))
for l in d['levels']: print(l['entered'], l['passed_min_eig'], l['cached_setups'], l['interior_setups'])
"
for w in 6 5; do
  timeout -k 10 300 python -u scratch/bench_variant.py scratch/lib/libvsf_w$w.so --no-cpu > gpurun_out/r8_bench_w$w.json 2> gpurun_out/r8_bench_w$w.err || { tail -5 gpurun_out/r8_bench_w$w.err; exit 1; }
  python -c "import json;d=json.load(open('gpurun_out/r8_bench_w$w.json'));print('waves $w', d['value'],d['kernel_ms']['lk'])"
done
