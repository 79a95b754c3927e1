#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/queues
rm -rf $OUT; mkdir -p $OUT
cd $R
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-batch-sweep --no-kernel-timing"
run() { name=$1; shift; env "$@" timeout 300 $B > $OUT/$name.log 2>&1; echo $name $(grep -o '"value": [0-9.]*' $OUT/$name.log | head -1); }
for q in 5 6 7 8 10 12 16; do run q$q GPU_MAX_HW_QUEUES=$q; done
run q8_b GPU_MAX_HW_QUEUES=8

