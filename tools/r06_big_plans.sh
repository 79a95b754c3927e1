#!/bin/bash
# GPU box (round 6): which contraction shapes should take the 256 x 128 kernel now that it runs the f16x3 arithmetic (small tiles vs forced big at several K splits)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r06
PROBE_BATCH=16 PROBE_KS="1,2,3,4,6" PROBE_MINK=128 timeout 900 python tools/big_gemm_probe.py > gpurun_out/r06/big_probe_f16_b16.txt 2>&1
PROBE_BATCH=1 PROBE_KS="1,2,3,4,6,8" PROBE_MINK=128 timeout 900 python tools/big_gemm_probe.py > gpurun_out/r06/big_probe_f16_b1.txt 2>&1
tail -45 gpurun_out/r06/big_probe_f16_b16.txt | cut -c1-220
tail -30 gpurun_out/r06/big_probe_f16_b1.txt | cut -c1-220
