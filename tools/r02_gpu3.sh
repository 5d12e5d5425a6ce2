#!/bin/bash
# chain + multi tests, full parity suite, then the restructured bench (plain + under kernel-trace)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
python -m pytest tests/test_gpu_chain.py -x -q -m gpu > gpurun_out/r02/chain_tests.log 2>&1; rc=$?
tail -25 gpurun_out/r02/chain_tests.log
[ $rc -eq 0 ] || exit $rc
python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r02/parity_tests.log 2>&1; rc=$?
tail -8 gpurun_out/r02/parity_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python bench.py > gpurun_out/r02/bench_a.json 2> gpurun_out/r02/bench_a.err; rc=$?
tail -c 3000 gpurun_out/r02/bench_a.json; tail -5 gpurun_out/r02/bench_a.err
exit $rc
