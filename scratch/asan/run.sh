#!/bin/bash
# GPU box: the C++ mirror driver and the library, host code under AddressSanitizer (device code as always)
set -u
cd "$GRAFT_REPO_ROOT"
python3 - <<'PY'
import sys
sys.path.insert(0, '.')
from _pkg import load_synth
synth = load_synth()
fr, _ = synth.sequence(77, 400, 300, 5)
fr.tofile('gpurun_out/asan_frames.raw')
PY
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
for args in "3 3 180 5.0 30 0.01 21 3 1.0" "0 0 120 9.0 30 0.01 21 3 1.0" "0 1 120 9.0 30 0.01 21 3 1.0" "3 3 180 5.0 0 0 0 0 0" "0 0 120 9.0 0 0 0 0 0"; do
  timeout -k 10 120 scratch/asan/frontend_mirror_asan gpurun_out/asan_frames.raw 400 300 5 gpurun_out/asan_out.txt $args > gpurun_out/asan_run.log 2>&1
  rc=$?
  echo "args [$args] rc $rc lines $(wc -l < gpurun_out/asan_out.txt)"
  if [ $rc -ne 0 ]; then head -40 gpurun_out/asan_run.log | cut -c1-220; exit 1; fi
done
timeout -k 10 120 scratch/asan/abi_exercise_asan gpurun_out/asan_frames.raw 400 300 5 > gpurun_out/asan_abi.log 2>&1
rc=$?
echo "abi exercise rc $rc"; tail -5 gpurun_out/asan_abi.log | cut -c1-250
[ $rc -eq 0 ] || { head -60 gpurun_out/asan_abi.log | cut -c1-250; exit 1; }
python3 - <<'PY'
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from _pkg import load_synth
from test_euroc_ingest import make_dataset
import shutil, os
root = 'gpurun_out/asan_euroc/V1_01_easy'
shutil.rmtree('gpurun_out/asan_euroc', ignore_errors=True)
fr, _ = load_synth().sequence(81, 376, 240, 12)
make_dataset(root, fr)
PY
timeout -k 10 120 scratch/asan/euroc_example_asan gpurun_out/asan_euroc/V1_01_easy gpurun_out/asan_euroc/V1_01_easy/calib.json gpurun_out/asan_euroc_out.txt > gpurun_out/asan_euroc.log 2>&1
rc=$?
echo "euroc example rc $rc lines $(wc -l < gpurun_out/asan_euroc_out.txt)"
[ $rc -eq 0 ] || { head -60 gpurun_out/asan_euroc.log | cut -c1-250; exit 1; }
