#!/bin/bash
for so in "$@"; do
  timeout -k 10 100 python scratch/bench_variant.py $so --config C5 --no-cpu --no-configs > gpurun_out/var.json 2> gpurun_out/var.err || { echo "$so FAILED"; tail -3 gpurun_out/var.err; continue; }
  python3 - "$so" <<'PY'
import json,sys
d=json.load(open("gpurun_out/var.json"))
k=d["kernel_ms"]
print("C5 %-16s value %.0f ms/step %.4f" % (sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"]), {a: round(b,4) for a,b in k.items()})
PY
done
