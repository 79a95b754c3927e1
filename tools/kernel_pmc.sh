#!/bin/bash
# PMC passes over ONE entry point's kernels (GPU box, via gpurun).  usage: tools/kernel_pmc.sh <entry> <kernel-substr> <tag> "<set1>" "<set2>" ...
# each set = space separated counters collected in one pass; output gpurun_out/pmc_<tag>.md
set -u
ENTRY=$1; SUB=$2; TAG=$3; shift 3
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$TAG.md
mkdir -p $R/gpurun_out; : > $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pk_$i
  rocprofv3 --kernel-trace --pmc $SET -d /tmp/pk_$i -o x -- python $R/tools/replay_kernel.py --kernel $ENTRY --reps 3 ${REPLAY_ARGS:-} > /tmp/pk_$i.log 2>&1
  DB=$(find /tmp/pk_$i -name '*_results.db' | head -1)
  if [ -z "$DB" ]; then echo "pass $i ($SET): no db" >> $OUT; tail -5 /tmp/pk_$i.log >> $OUT; continue; fi
  echo "## pass $i: $SET" >> $OUT
  python $R/tools/rocpd_pmc_summary.py $DB | grep -E "^\| kernel|^\|---|$SUB" >> $OUT
  rm -rf /tmp/pk_$i
done
cat $OUT
