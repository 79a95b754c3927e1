cd $GRAFT_REPO_ROOT
export COFI_GEMM=bf16x6
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "big_tiles" 2>&1 | tail -2
for k in 128 512 1024 3072; do python tools/gemm_one.py --shape 40960x1024x$k --kernel big --ks 1 --time 2>/dev/null; done
python tools/gemm_shapes.py --batch 16 > gpurun_out/r05/shapes_b16_d.txt 2>/dev/null; tail -n 1 gpurun_out/r05/shapes_b16_d.txt
