#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_batch.py tests/test_gpu_batch_c5.py tests/test_zz_no_alloc.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
timeout -k 10 300 python -u bench.py --no-cpu --occlusion 0.06 > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err || { tail -5 gpurun_out/r5_bench.err; exit 1; }
python -c "import json;d=json.load(open('gpurun_out/r5_bench.json'));print(d['value'],d['config']['verified_fraction'],d['kernel_ms'])"
