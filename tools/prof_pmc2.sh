#!/bin/bash
# PMC passes for every bench configuration (one rocprofv3 run per counter group, --pmc only, never combined with tracing),
# output under gpurun_out/pmc2/<config>/<group>/, then tools/pmc_collect.py -> gpurun_out/$ROUND_TAG_pmc.json (copy to profiles/).
# usage: prof_pmc2.sh [config keys...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
KEYS=${@:-cfg2_min_sum cfg3_sum_product cfg4_layered_m512 cfg5_qam16_min_sum f1_integer_min_sum f2_tasp_m126 f2_bp_m64 f2_asp_m64 exact_replay_generator}
run() { # key group counters...
  local key=$1 name=$2; shift 2
  timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc2/$key/$name -- python3 tools/pmc_workload.py $key > gpurun_out/pmc2/${key}_$name.log 2>&1 || { echo "pass $key/$name failed"; tail -5 gpurun_out/pmc2/${key}_$name.log; return 1; }
}
mkdir -p gpurun_out/pmc2
for key in $KEYS; do
  if [ "$key" = exact_replay_generator ]; then
    run $key sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM || exit 1
    run $key fetch FETCH_SIZE || exit 1
    run $key write WRITE_SIZE || exit 1
    echo "done $key"; continue
  fi
  run $key sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA || exit 1
  run $key sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM || exit 1
  run $key sq3 SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_LDS_ATOMIC || exit 1
  run $key fetch FETCH_SIZE || exit 1
  run $key write WRITE_SIZE || exit 1
  echo "done $key"
done
python3 tools/pmc_collect.py
