#!/bin/bash
# rocprofv3 PMC passes over ONE attention launch shape (GPU box, via gpurun).
# usage: tools/attn_pmc.sh <tag> <frames> <L> <S> <bf16x6|f32>
set -u
TAG=$1; FR=$2; L=$3; S=$4; AR=$5
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05/pmc_attn_$TAG.md
mkdir -p $R/gpurun_out/r05; : > $OUT
cd /tmp && export TMPDIR=/tmp
i=0
LIST=( "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
       "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16" \
       "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS SQ_WAVES" )
for SET in "${LIST[@]}"; do
  i=$((i+1))
  rm -rf /tmp/ap_$i
  timeout 180 rocprofv3 --kernel-trace --pmc $SET -d /tmp/ap_$i -o x -- python $R/tools/attn_one.py --frames $FR --L $L --S $S --arith $AR --reps 3 > /tmp/ap_$i.log 2>&1
  DB=$(find /tmp/ap_$i -name '*_results.db' | head -1)
  if [ -z "$DB" ]; then echo "pass $i ($SET): no db" >> $OUT; tail -3 /tmp/ap_$i.log >> $OUT; continue; fi
  echo "## pass $i: $SET" >> $OUT
  python $R/tools/rocpd_pmc_summary.py $DB | grep -E "attention_" >> $OUT
  rm -rf /tmp/ap_$i
done
cat $OUT
