#!/bin/bash
# GPU box: the short C2 bench line against several builds of the library (scratch/lib/*.so given as arguments)
for so in "$@"; do
  timeout -k 10 200 python scratch/bench_variant.py $so --no-cpu --no-configs > gpurun_out/var.json 2> gpurun_out/var.err || { echo "$so FAILED"; tail -3 gpurun_out/var.err; continue; }
  python3 - "$so" <<'PY'
import json,sys
d=json.load(open("gpurun_out/var.json"))
k=d["kernel_ms"]; c=d["roofline"]["sclk_mhz_under_load"]
print("%-28s value %.0f ms/step %.3f lk %.3f score %.3f sclk %.0f  lk_kcycles %.0f score_kcycles %.0f" % (sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], k["lk"], k["score"], c, k["lk"]*c, k["score"]*c))
PY
done
