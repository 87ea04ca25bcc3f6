#!/bin/bash
# Is the PMC crash ours?  (a) torch only, (b) torch + our library loaded, under rocprofv3 --pmc.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cat > /tmp/t_only.py <<'PY'
import torch
a = torch.rand(1024, 1024, device="cuda"); b = (a - 0.5).sum(); torch.cuda.synchronize(); print("torch ok", float(b))
PY
cat > /tmp/t_lib.py <<'PY'
import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
from _pkg import load_package
vsf = load_package(); vsf.lib()
import torch
a = torch.rand(1024, 1024, device="cuda"); b = (a - 0.5).sum(); torch.cuda.synchronize(); print("torch+lib ok", float(b))
PY
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES --output-format csv -d /tmp/pp1 -- python3 /tmp/t_only.py > gpurun_out/probe1.log 2>&1; echo "torch only rc=$?"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES --output-format csv -d /tmp/pp2 -- python3 /tmp/t_lib.py > gpurun_out/probe2.log 2>&1; echo "torch+lib rc=$?"
grep -h "ok\|SIGSEGV" gpurun_out/probe1.log gpurun_out/probe2.log
