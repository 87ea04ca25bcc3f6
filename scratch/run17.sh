#!/bin/bash
# full GPU suite, then C2 / C3 / C4 bench lines with the in-tree library
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread > gpurun_out/r17_tests.txt 2>&1 || { tail -30 gpurun_out/r17_tests.txt; exit 1; }
tail -2 gpurun_out/r17_tests.txt
for cfg in C2 C3 C4 C5; do
  timeout -k 10 300 python -u bench.py --no-cpu --config $cfg > gpurun_out/r17_bench_$cfg.json 2> gpurun_out/r17_bench_$cfg.err || { tail -5 gpurun_out/r17_bench_$cfg.err; exit 1; }
  python -c "import json;d=json.load(open('gpurun_out/r17_bench_$cfg.json'));print('$cfg', round(d['value']), round(d['ms_per_step'],3), {k:round(x,3) for k,x in d['kernel_ms'].items()})"
done
