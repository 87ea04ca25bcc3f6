#!/bin/bash
# full GPU suite, then the C2 bench line
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r9_tests.txt 2>&1 || { tail -30 gpurun_out/r9_tests.txt; exit 1; }
tail -2 gpurun_out/r9_tests.txt
timeout -k 10 300 python -u bench.py --no-cpu --occlusion 0.06 > gpurun_out/r9_bench.json 2> gpurun_out/r9_bench.err || { tail -5 gpurun_out/r9_bench.err; exit 1; }
python -c "import json;d=json.load(open('gpurun_out/r9_bench.json'));print(d['value'],d['config']['verified_fraction'],d['kernel_ms'])"
