#!/bin/bash
# GPU box (round 6): f16x3 plan table + fragment-order tail weights: tests, then the pipeline A/B (batch 16 and batch 1)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_forward_gpu.py -q -x -k "tail or chain or loftr or f16x3 or big_tiles or conv or kitti or golden or batch16 or stress" 2>&1 | tail -15 > gpurun_out/r06/t_step2.txt
tail -15 gpurun_out/r06/t_step2.txt
BENCH_ARGS="--batch 1" ROUNDS=2 STEPS=300 tools/ab_env.sh "COFI_TAIL_FRAG=0" "COFI_TAIL_FRAG=1" > gpurun_out/r06/ab_tail_frag.txt 2>&1
cat gpurun_out/r06/ab_tail_frag.txt
ROUNDS=2 STEPS=30 tools/ab_env.sh "COFI_F16X3=1" > gpurun_out/r06/ab_plans.txt 2>&1
cat gpurun_out/r06/ab_plans.txt
