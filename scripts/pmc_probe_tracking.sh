#!/bin/bash
# SQ counters of the headline kernel in the TRACKING regime (kinematic bounds and Jacobians, error_scale 0.02): the same
# passes as scripts/pmc_probe.sh, around scripts/ab_variants.py (AB_BOUNDS=tracking) on the shipped library.
#   bash scripts/pmc_probe_tracking.sh [outdir]
export TMPDIR=/tmp
OUT=${1:-gpurun_out/pmc_tracking}
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_FMA_F64 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS_F64 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  AB_BOUNDS=tracking rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python scripts/ab_variants.py pink_amd/csrc/libpinkhip.so > /dev/null 2> $OUT/p$i.err
  tail -1 $OUT/p$i.err
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
