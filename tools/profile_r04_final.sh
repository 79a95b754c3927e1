#!/bin/bash
# Run on the GPU box (via gpurun): kernel traces of the round's LAST build at the headline configuration (bf16x6, stack-mode batches of 16):
# one submission in flight (kernel durations not stretched by co-running kernels: the figure bench.py's `roofline` leg measures with HIP
# events) and four in flight (the timed loop itself).  usage: tools/profile_r04_final.sh   outputs under gpurun_out/prof_r04_final/
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r04_final
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
run_trace batch16_inflight1 "--inflight 1 --steps 6 --warmup 2"
run_trace batch16_inflight4 "--steps 8 --warmup 2"
grep -h '"value"' $OUT/*.log | cut -c1-160
ls -la $OUT
