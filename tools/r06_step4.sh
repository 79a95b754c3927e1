#!/bin/bash
# GPU box (round 6): f16x3 kernel reading pre-split weights (COFI_GEMM_W_F16PRE): tests, per-shape probe, pipeline A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_forward_gpu.py -q -x -k "f16x3 or big_tiles or kitti or golden or batch16 or stress or conv" 2>&1 | tail -15 > gpurun_out/r06/t_step4.txt
tail -15 gpurun_out/r06/t_step4.txt
COFI_F16X3_WPRE=1 timeout 600 python tools/f16_probe.py > gpurun_out/r06/f16_probe_wpre.txt 2>&1
grep -v "SAME BITS" gpurun_out/r06/f16_probe_wpre.txt | tail -24
ROUNDS=2 STEPS=30 tools/ab_env.sh "COFI_F16X3_WPRE=0" "COFI_F16X3_WPRE=1" > gpurun_out/r06/ab_wpre.txt 2>&1
cat gpurun_out/r06/ab_wpre.txt
