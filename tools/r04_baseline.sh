#!/bin/bash
# GPU box: round-4 baseline in the fp32-grade arithmetic (bf16x6): rates under a few switches + kernel traces.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_base
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-kernel-timing --no-batch-sweep --no-f32"
rate() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'f/s')"; }
run() { name=$1; shift; echo "== $name: $*"; env "$@" 2>$OUT/$name.err | tee $OUT/$name.json | rate; }
run x6 X=1 $B --gemm bf16x6
run x3 X=1 $B --gemm bf16x3
run x6_nodead COFI_DEAD_MAPS=0 $B --gemm bf16x6
run x3_nodead COFI_DEAD_MAPS=0 $B --gemm bf16x3
run x6_b2s4 X=1 $B --gemm bf16x6 --batch 2 --inflight 4 --steps 100
run x6_b4s4 X=1 $B --gemm bf16x6 --batch 4 --inflight 4 --steps 50
run x6_b4s2 X=1 $B --gemm bf16x6 --batch 4 --inflight 2 --steps 50
run x6_b16s2 X=1 $B --gemm bf16x6 --batch 16 --inflight 2 --steps 16
run x3_b2s4 X=1 $B --gemm bf16x3 --batch 2 --inflight 4 --steps 100
BENCH="$B --gemm bf16x6 --steps 20 --warmup 4 --repeats 1"
for cfg in "inflight4:" "inflight1:--inflight 1"; do
  name=${cfg%%:*}; extra=${cfg#*:}
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o x -- $BENCH $extra > $OUT/$name.log 2>&1
  DB=$(find /tmp/prof_$name -name '*_results.db' | head -1)
  python $R/tools/rocpd_summary.py $DB > $OUT/x6_${name}_kernel_trace.md 2>&1
  python $R/tools/rocpd_summary.py $DB --by-grid > $OUT/x6_${name}_by_grid.md 2>&1
  rm -rf /tmp/prof_$name
done
tail -3 $OUT/x6_inflight4_kernel_trace.md
