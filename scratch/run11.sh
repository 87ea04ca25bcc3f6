#!/bin/bash
# one SQ counter pass of the default bench with the in-tree library
set -u
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
rm -rf /tmp/prof_$tag && mkdir -p /tmp/prof_$tag
grp="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
timeout -s KILL 180 rocprofv3 --pmc $grp --output-format csv -d /tmp/prof_$tag/sq -- python3 bench.py --steps 3 --warmup 2 --no-cpu > $out/sq_${tag}.log 2>&1 || exit 1
python3 tools/pmc_summary.py /tmp/prof_$tag/sq $out/sq_${tag}.csv > /dev/null
grep "score_kernel" $out/sq_${tag}.csv
grp="SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD"
timeout -s KILL 180 rocprofv3 --pmc $grp --output-format csv -d /tmp/prof_$tag/sq2 -- python3 bench.py --steps 3 --warmup 2 --no-cpu > $out/sq2_${tag}.log 2>&1 || exit 1
python3 tools/pmc_summary.py /tmp/prof_$tag/sq2 $out/sq2_${tag}.csv > /dev/null
grep "score_kernel" $out/sq2_${tag}.csv
