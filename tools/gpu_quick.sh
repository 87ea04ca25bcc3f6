#!/bin/bash
# GPU box: the LK / batch parity tests, then a short C2 bench line (no CPU leg, no extra configs).
#   usage: bash tools/gpu_quick.sh <tag> [pytest -k expression]
set -u
tag=${1:-rXX}; kexpr=${2:-"lk or batch or parity or edge or no_alloc"}
out=gpurun_out
mkdir -p $out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread -k "$kexpr" > $out/${tag}_pytest.log 2>&1
rc=$?
tail -5 $out/${tag}_pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --no-cpu --no-configs > $out/${tag}_bench.json 2> $out/${tag}_bench.err
rc=$?
python3 - <<PY
import json
d=json.load(open("$out/${tag}_bench.json"))
print("value %.0f  ms/step %.3f  kernel_ms %s" % (d["value"], d["ms_per_step"], {k: round(v,3) for k,v in d["kernel_ms"].items()}))
c = d["roofline"]["sclk_mhz_under_load"]
print("sclk %.0f MHz   lk %.0f kcycles   score %.0f kcycles   step %.0f kcycles" % (c, d["kernel_ms"]["lk"] * c, d["kernel_ms"]["score"] * c, d["ms_per_step"] * c))
PY
exit $rc
