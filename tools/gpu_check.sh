#!/bin/bash
# GPU box: the whole GPU suite, then the default bench line (everything under its own time limit, chained).
#   usage: bash tools/gpu_check.sh <tag> [pytest args]
set -u
tag=${1:-rXX}; shift || true
out=gpurun_out
mkdir -p $out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread "$@" > $out/${tag}_pytest.log 2>&1
rc=$?
tail -15 $out/${tag}_pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
rc=$?
tail -3 $out/${tag}_bench.err
cut -c1-3000 $out/${tag}_bench.json
exit $rc
