#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -v --timeout 120 --timeout-method thread > gpurun_out/r3_pytest.log 2>&1
rc=$?
tail -25 gpurun_out/r3_pytest.log
[ $rc -ne 0 ] && exit $rc
for occ in 0 0.06; do
  timeout -k 10 300 python -u bench.py --no-cpu --occlusion $occ > gpurun_out/r3_bench_occ$occ.json 2> gpurun_out/r3_bench_occ$occ.err || { tail -5 gpurun_out/r3_bench_occ$occ.err; exit 1; }
  python -c "import json;d=json.load(open('gpurun_out/r3_bench_occ$occ.json'));print(d['value'],d['config']['verified_fraction'],d['kernel_ms'])"
done
timeout -k 10 300 python -u bench.py --no-cpu --detector anms --tracker lssd > gpurun_out/r3_bench_anms_lssd.json 2> gpurun_out/r3_bench_anms.err || { tail -5 gpurun_out/r3_bench_anms.err; exit 1; }
python -c "import json;d=json.load(open('gpurun_out/r3_bench_anms_lssd.json'));print(d['value'],d['config']['verified_fraction'],d['kernel_ms'])"
