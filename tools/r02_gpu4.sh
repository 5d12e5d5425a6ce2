#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
bash tools/prof_pmc2.sh > gpurun_out/r02/pmc2.log 2>&1; rc=$?
tail -12 gpurun_out/r02/pmc2.log
exit $rc
