#!/bin/bash
# round-2 measurement pass: profiles of the default bench, bench lines of the other configurations
set -o pipefail
tag=${1:-r02a}
out=gpurun_out
mkdir -p $out
bash tools/profile_round.sh $tag || { echo "profile_round failed"; exit 1; }
timeout -k 10 120 ./scratch/ubench/valu > $out/${tag}_valu_ubench.txt 2>&1 || exit 1
timeout -k 10 400 python3 -u bench.py > $out/${tag}_bench_C2.json 2> $out/${tag}_bench_C2.err || { tail -5 $out/${tag}_bench_C2.err; exit 1; }
echo "C2 done"
for cfg in C3 C4 C5; do
  timeout -k 10 600 python3 -u bench.py --config $cfg > $out/${tag}_bench_$cfg.json 2> $out/${tag}_bench_$cfg.err || { tail -5 $out/${tag}_bench_$cfg.err; exit 1; }
  echo "$cfg done"
done
timeout -k 10 300 python3 -u bench.py --no-cpu --tracker lssd > $out/${tag}_bench_lssd.json 2> $out/${tag}_bench_lssd.err || exit 1
timeout -k 10 300 python3 -u bench.py --no-cpu --detector anms --tracker lssd > $out/${tag}_bench_anms_lssd.json 2> $out/${tag}_bench_anms_lssd.err || exit 1
timeout -k 10 600 python3 -u bench.py --no-cpu --config C4 --detector anms > $out/${tag}_bench_C4_anms.json 2> $out/${tag}_bench_C4_anms.err || exit 1
timeout -k 10 300 python3 -u bench.py --no-cpu --occlusion 0 > $out/${tag}_bench_C2_noocc.json 2> $out/${tag}_bench_C2_noocc.err || exit 1
# --gpus 2 on a one-GPU box must fail loudly, not report n_gpus 1
timeout -k 10 120 python3 -u bench.py --gpus 2 --no-cpu > $out/${tag}_bench_gpus2.out 2> $out/${tag}_bench_gpus2.err; echo "gpus2 exit code $?" | tee $out/${tag}_bench_gpus2.rc
for f in C2 C3 C4 C5 lssd anms_lssd C4_anms C2_noocc; do python3 -c "
import json,sys
d=json.load(open('$out/${tag}_bench_$f.json'))
print('$f', round(d['value']), 'frames/s', round(d['ms_per_step'],3),'ms', 'verified', round(d['config']['verified_fraction'],4), 'lk', round(d['kernel_ms']['lk'],3), 'score', round(d['kernel_ms']['score'],3), 'frac', round(d['roofline']['frac'],4), 'cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('single_thread',{}).get('value'))
"; done
# per-kernel stats of the two ANMS lines
bash scratch/run21.sh || exit 1
cp $out/prof_c4a_kernel_stats.csv $out/${tag}_kernel_stats_C4_anms.csv
cp $out/prof_c2a_kernel_stats.csv $out/${tag}_kernel_stats_anms_lssd.csv
