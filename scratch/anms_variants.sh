#!/bin/bash
# GPU box: C2 step time with the ANMS detector's parameter variants (one box, one call).
set -u
out=gpurun_out
for v in "" "--anms-harris" "--anms-min-distance 5" "--anms-min-distance 20"; do
  echo "== anms $v"
  timeout -k 10 200 python bench.py --detector anms $v --no-cpu --no-configs --steps 10 --warmup 2 --preroll 40 > $out/anms_var.json 2> $out/anms_var.err || { tail -3 $out/anms_var.err; exit 1; }
  python - <<'PY'
import json
d=json.loads(open('gpurun_out/anms_var.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k: round(v,3) for k,v in d.get('kernel_ms',{}).items()} if 'kernel_ms' in d else d.get('config'))
PY
done
