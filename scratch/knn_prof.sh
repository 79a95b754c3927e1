#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/knn_prof; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_knn -o x -- python $R/scratch/knn_time.py > $OUT/run.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/prof_knn -name '*_results.db' | head -1) > $OUT/knn_kernel_trace.md 2>&1
head -30 $OUT/knn_kernel_trace.md | cut -c1-200
