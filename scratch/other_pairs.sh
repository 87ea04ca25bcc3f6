#!/bin/bash
# GPU box: the bench line for the other detector / tracker pairs (short runs, no CPU leg), stored for profiles/
set -u
out=gpurun_out
run() { name=$1; shift; timeout -k 10 300 python bench.py --no-cpu --no-configs "$@" > $out/r03c_bench_$name.json 2> $out/r03c_bench_$name.err || { echo "$name FAILED"; tail -3 $out/r03c_bench_$name.err; return 1; }
python3 - $name <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r03c_bench_%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d['value']), round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernel_ms'].items()}, d['config'].get('verified_fraction'))
PY
}
run lssd --tracker lssd && run anms --detector anms && run anms_lssd --detector anms --tracker lssd && run C4_anms --config C4 --detector anms --steps 10
