#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r02/all_gpu_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r02/all_gpu_tests.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
