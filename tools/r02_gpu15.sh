#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
python tools/time_bp_chain.py 2>&1 | grep "^BP"
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "bp or BP or sp_ or asp or tasp or golden" > gpurun_out/r02/sp_tests.log 2>&1; rc=$?
tail -4 gpurun_out/r02/sp_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 420 python tools/soak.py 200 31 --global > gpurun_out/r02/soak_global2.log 2>&1; echo "global soak rc $?"; grep -c " ok " gpurun_out/r02/soak_global2.log; grep -c MISMATCH gpurun_out/r02/soak_global2.log; grep "refused" gpurun_out/r02/soak_global2.log | head -3
