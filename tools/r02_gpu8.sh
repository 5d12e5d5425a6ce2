#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
timeout -k 10 200 tools/ab_flagship.bin 65536 0.0 > gpurun_out/r02/flagship_variants.txt 2>&1; cat gpurun_out/r02/flagship_variants.txt
timeout -k 10 200 tools/ab_flagship.bin 65536 2.0 >> gpurun_out/r02/flagship_variants.txt 2>&1; tail -4 gpurun_out/r02/flagship_variants.txt
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r02/all_gpu_tests.log 2>&1; rc=$?
tail -12 gpurun_out/r02/all_gpu_tests.log
exit $rc
