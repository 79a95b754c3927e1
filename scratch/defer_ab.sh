#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/defer_ab
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_forward_gpu.py -q -x 2>&1 | tail -5 > $OUT/tests.log
for m in 0 1; do
  COFI_DEFER_TAIL=$m timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-batch-sweep --no-kernel-timing > $OUT/b1_$m.log 2>&1
  COFI_DEFER_TAIL=$m timeout 300 python bench.py --batch 16 --steps 20 --warmup 4 --no-cpu-baseline --no-batch-sweep --no-kernel-timing > $OUT/b16_$m.log 2>&1
done
COFI_DEFER_TAIL=1 timeout 300 python bench.py --inflight 2 --steps 200 --warmup 20 --no-cpu-baseline --no-batch-sweep --no-kernel-timing > $OUT/b1_if2.log 2>&1
COFI_DEFER_TAIL=1 timeout 300 python bench.py --inflight 4 --steps 200 --warmup 20 --no-cpu-baseline --no-batch-sweep --no-kernel-timing > $OUT/b1_if4.log 2>&1
cat $OUT/tests.log
for f in $OUT/b*.log; do echo $f; grep -o '"value": [0-9.]*' $f | head -1; tail -2 $f | cut -c1-300 | grep -v '^{' ; done
