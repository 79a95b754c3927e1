#!/bin/bash
# rocprofv3 PMC passes over ONE dense bf16x6 contraction (GPU box, via gpurun).
# usage: tools/gemm_pmc.sh <tag> <shape MxNxK> <big|small> <ks> [kernel-name-substr]
set -u
TAG=$1; SHAPE=$2; KERN=$3; KS=$4; SUB=${5:-gemm_}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${PMC_DIR:-r05}/pmc_$TAG.md
mkdir -p $R/gpurun_out/${PMC_DIR:-r05}; : > $OUT
cd /tmp && export TMPDIR=/tmp
i=0
LIST=( "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
       "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16" \
       "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" )
for SET in "${LIST[@]}"; do
  i=$((i+1))
  rm -rf /tmp/gp_$i
  timeout 180 rocprofv3 --kernel-trace --pmc $SET -d /tmp/gp_$i -o x -- python $R/tools/gemm_one.py --shape $SHAPE --kernel $KERN --ks $KS --reps 3 ${GEMM_ONE_ARGS:-} > /tmp/gp_$i.log 2>&1
  DB=$(find /tmp/gp_$i -name '*_results.db' | head -1)
  if [ -z "$DB" ]; then echo "pass $i ($SET): no db" >> $OUT; tail -3 /tmp/gp_$i.log >> $OUT; continue; fi
  echo "## pass $i: $SET" >> $OUT
  python $R/tools/rocpd_pmc_summary.py $DB | grep -E "$SUB" >> $OUT
  rm -rf /tmp/gp_$i
done
cat $OUT
