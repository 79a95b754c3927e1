#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + separate PMC passes of the default bench configuration.
# Only the markdown summaries are kept (the rocpd databases are too large to travel back).
# usage: tools/profile_round.sh <tag>     outputs under gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 20 --warmup 4 --repeats 1 --no-f32 --no-cpu-baseline --no-kernel-timing --no-batch-sweep"
run_trace() {  # name, extra bench args
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$1 -o x -- $BENCH $2 > $OUT/$1.log 2>&1
  DB=$(find /tmp/prof_$1 -name '*_results.db' | head -1)
  python $R/tools/rocpd_summary.py $DB > $OUT/$1_kernel_trace.md 2>&1
  python $R/tools/rocpd_summary.py $DB --by-grid > $OUT/$1_by_grid.md 2>&1
  rm -rf /tmp/prof_$1
}
run_pmc() {  # name, counters
  rocprofv3 --kernel-trace --pmc $2 -d /tmp/prof_$1 -o x -- $BENCH --inflight 1 > $OUT/$1.log 2>&1
  python $R/tools/rocpd_pmc_summary.py $(find /tmp/prof_$1 -name '*_results.db' | head -1) > $OUT/$1_pmc.md 2>&1
  rm -rf /tmp/prof_$1
}
run_pmc_b16() {  # name, counters: BASELINE configs[2], 16 frames stacked per submission
  rocprofv3 --kernel-trace --pmc $2 -d /tmp/prof_$1 -o x -- $BENCH --inflight 1 --batch 16 --steps 4 --warmup 2 > $OUT/$1.log 2>&1
  python $R/tools/rocpd_pmc_summary.py $(find /tmp/prof_$1 -name '*_results.db' | head -1) > $OUT/$1_pmc.md 2>&1
  rm -rf /tmp/prof_$1
}
run_trace inflight4 ""
run_trace inflight1 "--inflight 1"
run_trace batch16 "--inflight 1 --batch 16 --steps 4 --warmup 2"
run_pmc fetch "FETCH_SIZE"
run_pmc write "WRITE_SIZE"
run_pmc mfma "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
run_pmc_b16 b16_fetch "FETCH_SIZE"
run_pmc_b16 b16_write "WRITE_SIZE"
run_pmc_b16 b16_mfma "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
grep -h '"value"' $OUT/*.log | cut -c1-160
ls -la $OUT
