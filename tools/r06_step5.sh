#!/bin/bash
# GPU box (round 6): one-instruction-per-half fp16 split: tests, probe, per-shape table of the whole contraction family, pipeline
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_ops_gpu.py -q -x -k "f16x3 or big_tiles" 2>&1 | tail -8 > gpurun_out/r06/t_step5.txt
tail -8 gpurun_out/r06/t_step5.txt
timeout 600 python tools/f16_probe.py > gpurun_out/r06/f16_probe_mix.txt 2>&1
grep -v "SAME BITS" gpurun_out/r06/f16_probe_mix.txt | tail -8
COFI_GEMM=bf16x6 timeout 600 python tools/gemm_shapes.py --batch 16 > gpurun_out/r06/gemm_shapes_batch16.txt 2>&1
COFI_GEMM=bf16x6 timeout 600 python tools/gemm_shapes.py --batch 1 > gpurun_out/r06/gemm_shapes_batch1.txt 2>&1
head -50 gpurun_out/r06/gemm_shapes_batch16.txt | cut -c1-120
ROUNDS=2 STEPS=30 tools/ab_env.sh "COFI_F16X3=1" > gpurun_out/r06/ab_mix.txt 2>&1
cat gpurun_out/r06/ab_mix.txt
