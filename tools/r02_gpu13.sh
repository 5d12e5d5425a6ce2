#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
: > gpurun_out/r02/ims_variants.txt
for r in 1 2; do for v in pack0 pack1 pack1_w2; do timeout -k 10 120 tools/ab_ims_$v.bin 65536 0.0 $v >> gpurun_out/r02/ims_variants.txt 2>&1; done; done
for v in pack0 pack1; do timeout -k 10 120 tools/ab_ims_$v.bin 65536 2.0 $v >> gpurun_out/r02/ims_variants.txt 2>&1; done
cat gpurun_out/r02/ims_variants.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "ims or IMS or integer or golden or ldpc_sim or soak or tier" > gpurun_out/r02/ims_tests.log 2>&1; rc=$?
tail -6 gpurun_out/r02/ims_tests.log
exit $rc
