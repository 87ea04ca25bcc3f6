#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/c1p
timeout -k 10 300 rocprofv3 --kernel-trace --hip-trace --stats --output-format csv -d /tmp/c1p -- python3 tools/run_euroc.py --synthetic 100 > gpurun_out/c1p.log 2>&1 || exit 1
f=$(find /tmp/c1p -name '*kernel_stats.csv' | head -1); cut -c1-150 "$f" | head -25 > gpurun_out/c1p_kernels.csv
f=$(find /tmp/c1p -name '*hip_api_stats.csv' | head -1); head -25 "$f" > gpurun_out/c1p_hip.csv
tail -1 gpurun_out/c1p.log
