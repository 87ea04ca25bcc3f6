#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out; rm -rf /tmp/pmc9
timeout -s KILL 180 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d /tmp/pmc9/a -- python3 bench.py --steps 3 --warmup 2 --no-cpu > gpurun_out/r10_pmc.log 2>&1 || exit 1
python3 tools/pmc_summary.py /tmp/pmc9/a gpurun_out/r10_pmc_sq.csv | grep "lk_fb"
timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc9/b -- python3 bench.py --steps 3 --warmup 2 --no-cpu > gpurun_out/r10_pmc2.log 2>&1 || exit 1
python3 tools/pmc_summary.py /tmp/pmc9/b gpurun_out/r10_pmc_fetch.csv | grep "lk_fb"
timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc9/c -- python3 bench.py --steps 3 --warmup 2 --no-cpu > gpurun_out/r10_pmc3.log 2>&1 || exit 1
python3 tools/pmc_summary.py /tmp/pmc9/c gpurun_out/r10_pmc_write.csv | grep "lk_fb"
