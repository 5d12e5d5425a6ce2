#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
timeout -k 10 500 python tools/soak.py 200 33 --global > gpurun_out/r03/soak_global.log 2>&1; echo "global rc $?"; tail -1 gpurun_out/r03/soak_global.log
timeout -k 10 700 python tools/soak.py 60 31 > gpurun_out/r03/soak_ms.log 2>&1; echo "ms rc $?"; tail -1 gpurun_out/r03/soak_ms.log
timeout -k 10 700 python tools/soak.py 30 32 --sp > gpurun_out/r03/soak_sp.log 2>&1; echo "sp rc $?"; tail -1 gpurun_out/r03/soak_sp.log
grep -c " ok" gpurun_out/r03/soak_global.log gpurun_out/r03/soak_ms.log gpurun_out/r03/soak_sp.log; grep -h "MISMATCH" gpurun_out/r03/soak_*.log | head
timeout -k 10 300 python tools/soak_mt.py 40 7 > gpurun_out/r03/soak_mt.log 2>&1; echo "mt rc $?"; tail -2 gpurun_out/r03/soak_mt.log
