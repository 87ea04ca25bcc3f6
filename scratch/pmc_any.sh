#!/bin/bash
# usage: bash scratch/pmc_any.sh <name> "<counters>" <bench args...>   (runs on the GPU box)
set -u
name=$1; ctr=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pl_$name && mkdir -p /tmp/pl_$name
timeout -s KILL 180 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pl_$name -- python3 bench.py --steps 3 --warmup 2 --no-cpu "$@" > gpurun_out/pmc_$name.log 2>&1 || { tail -5 gpurun_out/pmc_$name.log | cut -c1-200; exit 1; }
python3 tools/pmc_summary.py /tmp/pl_$name gpurun_out/pmc_$name.csv | grep "vsf::" | cut -c1-60,200-
