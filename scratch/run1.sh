#!/bin/bash
# first GPU pass of round 2: tests, VALU microbenchmark, LK statistics, occlusion sweep of the bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r1_pytest.log 2>&1
echo "pytest exit $?" | tee -a gpurun_out/r1_pytest.log
tail -5 gpurun_out/r1_pytest.log
timeout -k 10 120 ./scratch/ubench/valu > gpurun_out/r1_valu.log 2>&1 || exit 1
cat gpurun_out/r1_valu.log
for occ in 0 0.06; do
  timeout -k 10 200 python -u scratch/lk_stats.py 256 $occ > gpurun_out/r1_lkstats_$occ.json 2> gpurun_out/r1_lkstats_$occ.err || { tail -5 gpurun_out/r1_lkstats_$occ.err; exit 1; }
done
for occ in 0 0.03 0.06 0.1; do
  timeout -k 10 300 python -u bench.py --no-cpu --occlusion $occ > gpurun_out/r1_bench_occ$occ.json 2> gpurun_out/r1_bench_occ$occ.err || { tail -5 gpurun_out/r1_bench_occ$occ.err; exit 1; }
  cat gpurun_out/r1_bench_occ$occ.json
done
