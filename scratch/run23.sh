#!/bin/bash
# suite, then C2 / C3 / C4 / C5 bench lines without the CPU legs
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r23_tests.txt 2>&1 || { tail -30 gpurun_out/r23_tests.txt; exit 1; }
tail -2 gpurun_out/r23_tests.txt
for cfg in C2 C3 C4 C5; do
  timeout -k 10 400 python3 -u bench.py --no-cpu --config $cfg > gpurun_out/r23_$cfg.json 2> gpurun_out/r23_$cfg.err || { tail -5 gpurun_out/r23_$cfg.err; exit 1; }
  python3 -c "
import json
d=json.load(open('gpurun_out/r23_$cfg.json'))
print('$cfg', round(d['value']), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms'].items()})
"
done
