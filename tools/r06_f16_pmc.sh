#!/bin/bash
# GPU box (round 6): PMC passes over the f16x3 kernel on one large contraction, eight-wave (default) and four-wave geometry
cd $GRAFT_REPO_ROOT
export PMC_DIR=r06
tools/gemm_pmc.sh f16_w8_40960x1024x3072 40960x1024x3072 big 1 gemm_f16 > /dev/null 2>&1
GEMM_ONE_ARGS="--dbg 256" tools/gemm_pmc.sh f16_w4_40960x1024x3072 40960x1024x3072 big 1 gemm_f16 > /dev/null 2>&1
python tools/gemm_one.py --shape 40960x1024x3072 --kernel big --ks 1 --time
python tools/gemm_one.py --shape 40960x1024x3072 --kernel big --ks 1 --time --dbg 256
for f in gpurun_out/r06/pmc_f16_w8_40960x1024x3072.md gpurun_out/r06/pmc_f16_w4_40960x1024x3072.md; do echo "== $f"; grep -v "ROBUST\|Lb1E" $f | cut -c1-40,130-250 | head -60; done
