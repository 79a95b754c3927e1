#!/bin/bash
# GPU box (round 6): PMC passes over the f16x3 kernel in its shipped form (pre-split weights, 128 x 128 tiles two per CU) and the 256-row form
cd $GRAFT_REPO_ROOT
export PMC_DIR=r06
GEMM_ONE_ARGS="--presplit" tools/gemm_pmc.sh f16_bm128_wpre_40960x1024x3072 40960x1024x3072 big 1 gemm_f16 > /dev/null 2>&1
GEMM_ONE_ARGS="--presplit --dbg 512" tools/gemm_pmc.sh f16_bm256_wpre_40960x1024x3072 40960x1024x3072 big 1 gemm_f16 > /dev/null 2>&1
python tools/gemm_one.py --shape 40960x1024x3072 --kernel big --ks 1 --time --presplit
python tools/gemm_one.py --shape 40960x1024x3072 --kernel big --ks 1 --time --presplit --dbg 512
python tools/gemm_one.py --shape 40960x1024x3072 --kernel big --ks 1 --time
