#!/bin/bash
# randomised parity sweeps (single-stream API and batched API) against the oracle with the in-tree library
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -u scratch/fuzz_parity.py 240 ${1:-11} > gpurun_out/fuzz_parity.txt 2>&1 || { tail -20 gpurun_out/fuzz_parity.txt; exit 1; }
tail -2 gpurun_out/fuzz_parity.txt
timeout -k 10 400 python -u scratch/fuzz_batch.py 240 ${1:-11} > gpurun_out/fuzz_batch.txt 2>&1 || { tail -20 gpurun_out/fuzz_batch.txt; exit 1; }
tail -2 gpurun_out/fuzz_batch.txt
