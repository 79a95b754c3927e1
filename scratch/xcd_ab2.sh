#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/xcd_ab2
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_forward_gpu.py -q -x 2>&1 | tail -3 > $OUT/tests.log
for m in 0 2; do
  COFI_GEMM_XCD=$m timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-batch-sweep > $OUT/b1_$m.log 2>&1
  COFI_GEMM_XCD=$m timeout 300 python bench.py --batch 16 --steps 20 --warmup 4 --no-cpu-baseline --no-batch-sweep > $OUT/b16_$m.log 2>&1
done
cat $OUT/tests.log
for f in $OUT/b*.log; do echo $f; python - $f <<'P'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); print(round(d["value"],1), d["ms_per_step"], d.get("kernel_ms_per_frame",{}).get("gemm"), d["roofline"]["frac"])
P
done
bash tools/profile_round.sh r01_m 2>&1 | tail -3
