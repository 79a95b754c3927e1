#!/bin/bash
# GPU box (round 6): f16x3 kernel, 128 x 128 tiles two per CU: tests, then the plan probes (forced K splits) at batch 16, batch 1 and the stress configuration
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_ops_gpu.py -q -x -k "f16x3 or big_tiles" 2>&1 | tail -6 > gpurun_out/r06/t_step7.txt
tail -6 gpurun_out/r06/t_step7.txt
PROBE_BATCH=16 PROBE_KS="1,2,3,4,6" PROBE_MINK=128 timeout 900 python tools/big_gemm_probe.py > gpurun_out/r06/big_probe_bm128_b16.txt 2>&1
PROBE_BATCH=1 PROBE_KS="1,2,3,4,6,8" PROBE_MINK=128 timeout 900 python tools/big_gemm_probe.py > gpurun_out/r06/big_probe_bm128_b1.txt 2>&1
PROBE_STRESS=1 PROBE_BATCH=1 PROBE_KS="1,2,3,4,6,8" PROBE_MINK=128 PROBE_MINM=2048 timeout 900 python tools/big_gemm_probe.py > gpurun_out/r06/big_probe_bm128_stress.txt 2>&1
tail -3 gpurun_out/r06/big_probe_bm128_b16.txt | cut -c1-200
tail -2 gpurun_out/r06/big_probe_bm128_b1.txt | cut -c1-200
tail -2 gpurun_out/r06/big_probe_bm128_stress.txt | cut -c1-200
