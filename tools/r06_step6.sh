#!/bin/bash
# GPU box (round 6): cvt_pk split + fp32 maximum tracking; attention image in column-major piece order; batch-1 norm finalize A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_ops_gpu.py -q -x -k "f16x3 or big_tiles or attention" 2>&1 | tail -8 > gpurun_out/r06/t_step6.txt
tail -8 gpurun_out/r06/t_step6.txt
timeout 600 python tools/f16_probe.py > gpurun_out/r06/f16_probe_max3.txt 2>&1
grep -v "SAME BITS" gpurun_out/r06/f16_probe_max3.txt | tail -3
timeout 600 python tools/attn_presplit_probe.py > gpurun_out/r06/attn_presplit_probe_b.txt 2>&1
cat gpurun_out/r06/attn_presplit_probe_b.txt
BENCH_ARGS="--batch 1" ROUNDS=2 STEPS=300 tools/ab_env.sh "COFI_NORM_FINALIZE_FRAMES=1" "COFI_NORM_FINALIZE_FRAMES=2" > gpurun_out/r06/ab_finalize_b1.txt 2>&1
cat gpurun_out/r06/ab_finalize_b1.txt
