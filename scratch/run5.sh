#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r6_pytest.log 2>&1; rc=$?
tail -4 gpurun_out/r6_pytest.log; [ $rc -ne 0 ] && { grep -n "Error\|assert" gpurun_out/r6_pytest.log | head -20; exit $rc; }
timeout -k 10 300 python -u tools/run_euroc.py --synthetic 150 > gpurun_out/r6_euroc.json 2> gpurun_out/r6_euroc.err || { tail -5 gpurun_out/r6_euroc.err; exit 1; }
cat gpurun_out/r6_euroc.json
