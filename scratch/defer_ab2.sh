#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/defer_ab2
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_forward_gpu.py -q -x 2>&1 | tail -5 > $OUT/tests.log
python -c "import torch; print(torch.cuda.Stream.priority_range())" > $OUT/prio.log 2>&1
B="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-batch-sweep --no-kernel-timing"
run() { name=$1; shift; env "$@" timeout 300 $B > $OUT/$name.log 2>&1; echo $name $(grep -o '"value": [0-9.]*' $OUT/$name.log | head -1); }
run base COFI_DEFER_TAIL=0
run shared_p0 COFI_DEFER_TAIL=1
run slot_p0 COFI_DEFER_TAIL=1 COFI_TAIL_STREAMS=slot
run shared_hi COFI_DEFER_TAIL=1 COFI_TAIL_PRIORITY=-1
run shared_lo COFI_DEFER_TAIL=1 COFI_TAIL_PRIORITY=1
run slot_lo COFI_DEFER_TAIL=1 COFI_TAIL_STREAMS=slot COFI_TAIL_PRIORITY=1
run shared_q4 COFI_DEFER_TAIL=1 GPU_MAX_HW_QUEUES=4
run shared_q16 COFI_DEFER_TAIL=1 GPU_MAX_HW_QUEUES=16
cat $OUT/tests.log $OUT/prio.log
