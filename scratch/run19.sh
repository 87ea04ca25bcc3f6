#!/bin/bash
# LDS counters of the LK kernel with the in-tree library
set -u
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
rm -rf /tmp/prof_$tag && mkdir -p /tmp/prof_$tag
grp="SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVES"
timeout -s KILL 180 rocprofv3 --pmc $grp --output-format csv -d /tmp/prof_$tag/sq -- python3 bench.py --steps 3 --warmup 2 --no-cpu > $out/lds_${tag}.log 2>&1 || exit 1
python3 tools/pmc_summary.py /tmp/prof_$tag/sq $out/lds_${tag}.csv > /dev/null
grep "lk_fb" $out/lds_${tag}.csv
