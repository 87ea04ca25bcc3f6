#!/bin/bash
for so in "$@"; do
  timeout -k 10 200 python scratch/bench_variant.py $so --no-cpu --no-configs > gpurun_out/var.json 2> gpurun_out/var.err || { echo "$so FAILED"; tail -3 gpurun_out/var.err; continue; }
  python3 - "$so" <<'PY'
import json,sys
d=json.load(open("gpurun_out/var.json"))
k=d["kernel_ms"]
print("%-20s value %.0f ms/step %.3f lk %.3f score %.3f select %.4f pyr %.3f" % (sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], k["lk"], k["score"], k["select"], k["pyramid"]))
PY
done
