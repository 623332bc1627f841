#!/bin/bash
# SQ counters of the solve kernel in one input regime:  bash scripts/pmc_regime.sh tracking [config]
export TMPDIR=/tmp
R=${1:-tracking}; C=${2:-draco3}
OUT=gpurun_out/pmc_$R
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python scripts/solve_regime_once.py $R $C > /dev/null 2> $OUT/p$i.err
done
OUT=$OUT python - <<'PY'
import csv, collections, glob, os
for f in sorted(glob.glob(os.environ['OUT']+'/p*/p_counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'ik_solve' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items(): print(f.split('/')[-2], k, len(v), sum(v)/len(v))
PY
