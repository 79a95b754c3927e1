#!/bin/bash
# GPU box (round 6): attention with K / V split once: tests, probe, pipeline A/B; plan-table A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_ops_gpu.py -q -x -k "attention" 2>&1 | tail -15 > gpurun_out/r06/t_step3.txt
tail -15 gpurun_out/r06/t_step3.txt
timeout 600 python tools/attn_presplit_probe.py > gpurun_out/r06/attn_presplit_probe.txt 2>&1
cat gpurun_out/r06/attn_presplit_probe.txt
ROUNDS=2 STEPS=30 tools/ab_env.sh "COFI_ATTN_PRESPLIT_ROWS=0" "COFI_ATTN_PRESPLIT_ROWS=4096" > gpurun_out/r06/ab_presplit.txt 2>&1
cat gpurun_out/r06/ab_presplit.txt
ROUNDS=2 STEPS=30 tools/ab_env.sh "COFI_GEMM_F16_TABLE=0" "COFI_GEMM_F16_TABLE=1" > gpurun_out/r06/ab_f16_table.txt 2>&1
cat gpurun_out/r06/ab_f16_table.txt
