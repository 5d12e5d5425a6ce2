#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
timeout -k 10 800 python -m pytest tests/test_gpu_shapes.py tests/test_gpu_chain.py -x -q -m gpu > gpurun_out/r02/new_tests.log 2>&1; rc=$?
tail -6 gpurun_out/r02/new_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ims or integer or unsupported" 2>&1 | tail -3
bash tools/r02_measure.sh
