#!/bin/bash
# round-2 GPU call 1: executed drop-in tests, FETCH/WRITE calibration, PMC passes of the flagship at HEAD
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "upstream_bp_simulation_symbol or exact_replay" > gpurun_out/r02/dropin_tests.log 2>&1 || { tail -30 gpurun_out/r02/dropin_tests.log; exit 1; }
tail -3 gpurun_out/r02/dropin_tests.log
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r02/cal_fetch -- tools/ubench_fetch.bin > gpurun_out/r02/cal_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r02/cal_write -- tools/ubench_fetch.bin > gpurun_out/r02/cal_write.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/cal_trace -- tools/ubench_fetch.bin > gpurun_out/r02/cal_trace.log 2>&1
python3 - <<'PY'
import csv, glob
for tag in ("cal_fetch", "cal_write"):
    for f in glob.glob(f"gpurun_out/r02/{tag}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            print(tag, row["Kernel_Name"][:40], row["Counter_Name"], row["Counter_Value"])
PY
bash tools/prof_pmc.sh 2 r02_head > gpurun_out/r02/pmc_head_summary.txt 2>&1
tail -40 gpurun_out/r02/pmc_head_summary.txt
