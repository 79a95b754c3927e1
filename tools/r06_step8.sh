#!/bin/bash
# GPU box (round 6): 128 x 128 two-per-CU form of the f16x3 kernel as the default: parity tests, the pipeline A/B against the 256-row form, batch 1 too
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_forward_gpu.py -q -x -k "f16x3 or big_tiles or kitti or golden or batch16 or stress or conv" 2>&1 | tail -6 > gpurun_out/r06/t_step8.txt
tail -6 gpurun_out/r06/t_step8.txt
ROUNDS=2 STEPS=30 tools/ab_env.sh "COFI_GEMM_F16_BM256=1" "COFI_GEMM_F16_BM256=0" > gpurun_out/r06/ab_bm128.txt 2>&1
cat gpurun_out/r06/ab_bm128.txt
BENCH_ARGS="--batch 1" ROUNDS=2 STEPS=300 tools/ab_env.sh "COFI_GEMM_F16_BM256=1" "COFI_GEMM_F16_BM256=0" > gpurun_out/r06/ab_bm128_b1.txt 2>&1
cat gpurun_out/r06/ab_bm128_b1.txt
