#!/bin/bash
# kernel stats of the ANMS bench lines: bash scratch/run21.sh
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
for cfg in c4a c2a; do
  rm -rf /tmp/prof_$cfg && mkdir -p /tmp/prof_$cfg
  if [ $cfg = c4a ]; then
    timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$cfg/stats -- python3 bench.py --no-cpu --config C4 --detector anms > $out/prof_${cfg}_bench.log 2>&1 || { tail -5 $out/prof_${cfg}_bench.log; exit 1; }
  else
    timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$cfg/stats -- python3 bench.py --no-cpu --detector anms --tracker lssd > $out/prof_${cfg}_bench.log 2>&1 || { tail -5 $out/prof_${cfg}_bench.log; exit 1; }
  fi
  f=$(find /tmp/prof_$cfg/stats -name '*kernel_stats.csv' | head -1)
  grep "^\"Name\"\|vsf::" "$f" > $out/prof_${cfg}_kernel_stats.csv
  python3 - $cfg <<'PY'
import csv, sys
print(sys.argv[1])
for r in csv.DictReader(open('gpurun_out/prof_%s_kernel_stats.csv' % sys.argv[1])):
    print(' ', r['Name'][:50].ljust(52), r['Calls'], round(float(r['AverageNs'])/1e6,4), 'ms', r['Percentage'])
PY
done
