#!/bin/bash
# Run on the GPU box (via gpurun): kernel traces + separate PMC passes of the round-4 bench configurations, all in the fp32-grade arithmetic
# (bf16x6): the headline (stack-mode batches of 16, 4 submissions in flight) and the batch-1 pipeline (BASELINE configs[1]).
# Only the markdown summaries are kept.  usage: tools/profile_r04.sh <tag>     outputs under gpurun_out/prof_<tag>/
set -u
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --repeats 1 --no-f32 --no-cpu-baseline --no-kernel-timing --no-batch-sweep --no-steady"
run_trace() {  # name, extra bench args
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$1 -o x -- $BENCH $2 > $OUT/$1.log 2>&1
  DB=$(find /tmp/prof_$1 -name '*_results.db' | head -1)
  python $R/tools/rocpd_summary.py $DB > $OUT/$1_kernel_trace.md 2>&1
  python $R/tools/rocpd_summary.py $DB --by-grid > $OUT/$1_by_grid.md 2>&1
  rm -rf /tmp/prof_$1
}
run_pmc() {  # name, counters, bench args
  rocprofv3 --kernel-trace --pmc $2 -d /tmp/prof_$1 -o x -- $BENCH $3 > $OUT/$1.log 2>&1
  python $R/tools/rocpd_pmc_summary.py $(find /tmp/prof_$1 -name '*_results.db' | head -1) > $OUT/$1_pmc.md 2>&1
  rm -rf /tmp/prof_$1
}
MF="SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
run_trace batch16 "--steps 8 --warmup 2"
run_trace batch1_inflight4 "--batch 1 --steps 20 --warmup 4"
run_trace batch1_inflight1 "--batch 1 --inflight 1 --steps 20 --warmup 4"
run_pmc b16_fetch "FETCH_SIZE" "--inflight 1 --steps 4 --warmup 2"
run_pmc b16_write "WRITE_SIZE" "--inflight 1 --steps 4 --warmup 2"
run_pmc b16_mfma "$MF" "--inflight 1 --steps 4 --warmup 2"
run_pmc b1_fetch "FETCH_SIZE" "--batch 1 --inflight 1 --steps 20 --warmup 4"
run_pmc b1_write "WRITE_SIZE" "--batch 1 --inflight 1 --steps 20 --warmup 4"
run_pmc b1_mfma "$MF" "--batch 1 --inflight 1 --steps 20 --warmup 4"
grep -h '"value"' $OUT/*.log | cut -c1-120
ls -la $OUT
