#!/bin/bash
# full GPU suite with the in-tree library, then C2 bench lines of the named variants
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r14_tests.txt 2>&1 || { tail -30 gpurun_out/r14_tests.txt; exit 1; }
tail -2 gpurun_out/r14_tests.txt
bash scratch/run13.sh "$@"
