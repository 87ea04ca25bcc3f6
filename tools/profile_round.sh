#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats of the default bench, then one PMC pass per
# counter group (each in its own run), summarised into gpurun_out/prof_<tag>_*.csv.
#   usage: bash tools/profile_round.sh <tag>
set -u
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
rm -rf /tmp/prof_$tag && mkdir -p /tmp/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag/stats -- python3 bench.py --no-cpu --no-configs > $out/prof_${tag}_bench.log 2>&1 || exit 1
f=$(find /tmp/prof_$tag/stats -name '*kernel_stats.csv' | head -1)
grep "^\"Name\"\|vsf::" "$f" > $out/prof_${tag}_kernel_stats.csv
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout -s KILL 180 rocprofv3 --pmc $grp --output-format csv -d /tmp/prof_$tag/$name -- python3 bench.py --steps 3 --warmup 2 --no-cpu --no-configs > $out/prof_${tag}_pmc_$name.log 2>&1 || exit 1
  python3 tools/pmc_summary.py /tmp/prof_$tag/$name $out/prof_${tag}_pmc_$name.csv | grep "vsf::" > /dev/null
  grep "^kernel\|vsf::" $out/prof_${tag}_pmc_$name.csv > $out/tmp.csv && mv $out/tmp.csv $out/prof_${tag}_pmc_$name.csv
done
grep "^{\"metric\"" $out/prof_${tag}_bench.log | tail -1
cat $out/prof_${tag}_kernel_stats.csv | cut -c1-200
