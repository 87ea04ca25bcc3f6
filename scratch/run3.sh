#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_anms.py tests/test_gpu_edge_cases.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
timeout -k 10 300 python -u bench.py --no-cpu --occlusion 0.06 > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.err || { tail -5 gpurun_out/r4_bench.err; exit 1; }
python -c "import json;d=json.load(open('gpurun_out/r4_bench.json'));print(d['value'],d['config']['verified_fraction'],d['kernel_ms'])"
timeout -k 10 300 python -u bench.py --no-cpu --detector anms --tracker lssd > gpurun_out/r4_bench_anms_lssd.json 2> gpurun_out/r4_bench_anms.err || { tail -5 gpurun_out/r4_bench_anms.err; exit 1; }
python -c "import json;d=json.load(open('gpurun_out/r4_bench_anms_lssd.json'));print(d['value'],d['config']['verified_fraction'],d['kernel_ms'])"
