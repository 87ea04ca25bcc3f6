#!/bin/bash
# like variants.sh, with the ANMS detector
for so in "$@"; do
  timeout -k 10 200 python scratch/bench_variant.py $so --detector anms --no-cpu --no-configs --steps 10 --preroll 40 > gpurun_out/var.json 2> gpurun_out/var.err || { echo "$so FAILED"; tail -3 gpurun_out/var.err; continue; }
  python3 - "$so" <<'PY'
import json,sys
d=json.load(open("gpurun_out/var.json"))
k=d["kernel_ms"]
print("ANMS %-24s value %.0f ms/step %.3f lk %.3f score %.3f select %.3f" % (sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], k["lk"], k["score"], k["select"]))
PY
done
