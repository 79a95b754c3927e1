#!/bin/bash
# rocprofv3 PMC passes over gemm_planes_kernel (GPU box, via gpurun).  usage: tools/planes_pmc.sh <tag> <shape> <cfg> <ks>
set -u
TAG=$1; SHAPE=$2; CFG=$3; KS=$4; SETS=${5:-all}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/planes_pmc_$TAG.md
mkdir -p $R/gpurun_out; : > $OUT
cd /tmp && export TMPDIR=/tmp
i=0
if [ "$SETS" = "sq" ]; then LIST=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE"); else LIST=( "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TA_BUSY_avr TCP_TA_DATA_STALL_CYCLES_sum" "FETCH_SIZE" "GRBM_GUI_ACTIVE"); fi
for SET in "${LIST[@]}"; do
  i=$((i+1))
  rm -rf /tmp/pp_$i
  timeout 120 rocprofv3 --kernel-trace --pmc $SET -d /tmp/pp_$i -o x -- python $R/tools/planes_probe.py --shape $SHAPE --cfg $CFG --ks $KS --reps 3 > /tmp/pp_$i.log 2>&1
  DB=$(find /tmp/pp_$i -name '*_results.db' | head -1)
  if [ -z "$DB" ]; then echo "pass $i ($SET): no db" >> $OUT; tail -3 /tmp/pp_$i.log >> $OUT; continue; fi
  echo "## pass $i: $SET" >> $OUT
  python $R/tools/rocpd_pmc_summary.py $DB | grep -E "gemm_planes" >> $OUT
  rm -rf /tmp/pp_$i
done
cat $OUT
