#!/bin/bash
# PMC probe of the solve kernel (default: draco3, B=65536): issue/stall breakdown.
#   bash scripts/pmc_probe.sh [outdir] [bench.py arguments, e.g. --config jvrc]
export TMPDIR=/tmp
OUT=${1:-gpurun_out/pmc}
shift
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --headline-only "$@" > /dev/null 2> $OUT/p$i.err
  tail -2 $OUT/p$i.err
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
