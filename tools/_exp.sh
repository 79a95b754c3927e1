cd $GRAFT_REPO_ROOT
for d in 0 4096 8192 12288 4 4100 8196 12292; do python tools/gemm_one.py --shape 40960x1024x3072 --kernel big --ks 1 --time --dbg $d; done
