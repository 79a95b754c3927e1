#!/bin/bash
# A/B of the GEMM tile order (COFI_GEMM_XCD 0 = hardware, 1 = contiguous, 2 = estimate): tests, bench, FETCH_SIZE pass.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/xcd_ab
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "gemm or conv" 2>&1 | tail -3 > $OUT/tests.log
for m in 0 1 2; do
  COFI_GEMM_XCD=$m timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-batch-sweep > $OUT/b1_$m.log 2>&1
  COFI_GEMM_XCD=$m timeout 300 python bench.py --batch 16 --steps 20 --warmup 4 --no-cpu-baseline --no-batch-sweep > $OUT/b16_$m.log 2>&1
done
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-batch-sweep --inflight 1"
for m in 0 2; do
  COFI_GEMM_XCD=$m timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_f$m -o x -- $BENCH > $OUT/fetch_$m.log 2>&1
  python $R/tools/rocpd_pmc_summary.py $(find /tmp/prof_f$m -name '*_results.db' | head -1) > $OUT/fetch_${m}_pmc.md 2>&1
  rm -rf /tmp/prof_f$m
done
cat $OUT/tests.log
for f in $OUT/b*.log; do echo $f; python - $f <<'P'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); print(round(d["value"],1), d["ms_per_step"], d.get("kernel_ms_per_frame",{}).get("gemm"), d["roofline"]["frac"])
P
done
head -12 $OUT/fetch_0_pmc.md; head -12 $OUT/fetch_2_pmc.md
