#!/bin/bash
# PMC passes over the min-sum decode (variant given as $1, default 2).  Each pass is its own rocprofv3 run with
# --pmc only (never combined with tracing), output under gpurun_out/pmc_<tag>/.
set -e
V=${1:-2}
TAG=${2:-v$V}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export AB_ROUNDS=2
run() { # name, counters...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_${TAG}/$name -- python3 tools/ab_ms.py $V > gpurun_out/pmc_${TAG}_$name.log 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM
run sq3 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_LDS_ATOMIC SQ_INSTS_BRANCH SQ_IFETCH SQ_WAVES GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 - <<'PY'
import csv, glob, collections, os
tag=os.environ.get('TAG_','')
import sys
for d in sorted(glob.glob('gpurun_out/pmc_*/*')):
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
        for row in csv.DictReader(open(f)):
            k=row['Kernel_Name']
            if 'ms_' not in k and 'flood' not in k: continue
            acc[k][row['Counter_Name']]+=float(row['Counter_Value'])
            n[(k,row['Counter_Name'])]+=1
        for k in acc:
            print(d.split('/')[-2], d.split('/')[-1], k[:60])
            for c,v in acc[k].items(): print('    %-28s per-dispatch %.4g  (dispatches %d)'%(c, v/n[(k,c)], n[(k,c)]))
PY
