#!/bin/bash
# PMC pass for the LSSD configuration (runs on the GPU box)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pl && mkdir -p /tmp/pl
timeout -s KILL 180 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM --output-format csv -d /tmp/pl/sq -- python3 bench.py --tracker lssd --steps 3 --warmup 2 --no-cpu > gpurun_out/pmc_lssd.log 2>&1 || exit 1
python3 tools/pmc_summary.py /tmp/pl/sq gpurun_out/pmc_lssd_sq.csv | grep "lssd"
