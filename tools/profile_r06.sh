#!/bin/bash
# Run on the GPU box (via gpurun): kernel traces + separate PMC passes of the round-6 bench configurations in the fp32-grade arithmetic
# (bf16x6): the headline (stack-mode batches of 16; one and four submissions in flight), the batch-1 pipeline (BASELINE configs[1]) and
# the stress configuration (configs[4]); FETCH_SIZE / WRITE_SIZE / MFMA passes -> pmc_traffic*.json.  Only the markdown summaries and the
# json files are kept.  usage: tools/profile_r06.sh <commit stamp>     outputs under gpurun_out/prof_r06/
set -u
STAMP=${1:-unknown}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r06
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
MF="SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
run_trace batch16_inflight1 "--inflight 1 --steps 6 --warmup 2"
run_trace batch16_inflight4 "--steps 8 --warmup 2"
run_trace batch1_inflight1 "--batch 1 --inflight 1 --steps 20 --warmup 4"
run_trace batch1_inflight4 "--batch 1 --steps 20 --warmup 4"
run_pmc b16_fetch "FETCH_SIZE" "--inflight 1 --steps 4 --warmup 2"
run_pmc b16_write "WRITE_SIZE" "--inflight 1 --steps 4 --warmup 2"
run_pmc b16_mfma "$MF" "--inflight 1 --steps 4 --warmup 2"
run_pmc b1_fetch "FETCH_SIZE" "--batch 1 --inflight 1 --steps 20 --warmup 4"
run_pmc b1_write "WRITE_SIZE" "--batch 1 --inflight 1 --steps 20 --warmup 4"
run_pmc b1_mfma "$MF" "--batch 1 --inflight 1 --steps 20 --warmup 4"
run_pmc stress_fetch "FETCH_SIZE" "--stress --batch 1 --inflight 1 --steps 6 --warmup 2"
run_pmc stress_write "WRITE_SIZE" "--stress --batch 1 --inflight 1 --steps 6 --warmup 2"
# HBM bytes per launch / per frame of the kernel families (FETCH doubled per the gfx950 note of MI355X_MICROARCH.md)
python $R/tools/pmc_to_json.py $OUT/b16_fetch_pmc.md $OUT/b16_write_pmc.md auto "$STAMP (round 6; bf16x6, stack-mode batches of 16, one submission in flight)" 12 16 > $OUT/pmc_traffic.json
python $R/tools/pmc_to_json.py $OUT/b1_fetch_pmc.md $OUT/b1_write_pmc.md auto "$STAMP (round 6; bf16x6, batch 1, one frame in flight)" 12 1 > $OUT/pmc_traffic_batch1.json
python $R/tools/pmc_to_json.py $OUT/stress_fetch_pmc.md $OUT/stress_write_pmc.md auto "$STAMP (round 6; bf16x6, stress configuration 896 x 1600 / 40960 points, one frame in flight)" 16 1 > $OUT/pmc_traffic_stress.json
grep -h '"value"' $OUT/*.log | cut -c1-120
ls -la $OUT
