#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_gpu_shapes.py -x -q -m gpu > gpurun_out/r02/shape_tests.log 2>&1; rc=$?
tail -6 gpurun_out/r02/shape_tests.log
[ $rc -eq 0 ] || exit $rc
rm -f gpurun_out/r02/time_after_w.log
for cfg in "0 64 2.0 32768" "0 64 0.0 8192" "3 64 0.0 65536" "3 126 0.0 32768" "3 512 0.0 4096" "8 64 0.0 65536" "8 512 0.0 4096" "3 32 3.0 262144"; do python tools/time_decoder.py $cfg >> gpurun_out/r02/time_after_w.log 2>&1; done
grep "^dec" gpurun_out/r02/time_after_w.log
