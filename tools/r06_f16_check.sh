#!/bin/bash
# GPU box (round 6): the f16x3 kernel - its tests, the per-shape probe on the forward's own operands, the whole-pipeline A/B, then the GPU suite
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "f16x3 or big_tiles" 2>&1 | tail -30 > gpurun_out/r06/t_f16.txt
tail -30 gpurun_out/r06/t_f16.txt
timeout 600 python tools/f16_probe.py > gpurun_out/r06/f16_probe.txt 2>&1
grep -v "SAME BITS" gpurun_out/r06/f16_probe.txt | tail -30
ROUNDS=2 STEPS=30 tools/ab_env.sh "COFI_F16X3=0" "COFI_F16X3=1" > gpurun_out/r06/ab_f16.txt 2>&1
cat gpurun_out/r06/ab_f16.txt
if [ "${FULL:-1}" = "1" ]; then timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -40 > gpurun_out/r06/t_gpu.txt; tail -40 gpurun_out/r06/t_gpu.txt; fi
