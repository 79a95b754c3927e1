cd $GRAFT_REPO_ROOT
export COFI_GEMM=bf16x6
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q 2>&1 | tail -4
for m in 0 1; do echo "inlaunch $m"; COFI_GEMM_INLAUNCH=$m python tools/gemm_shapes.py --batch 1 2>/dev/null | tail -n 1;  COFI_GEMM_INLAUNCH=$m python tools/gemm_shapes.py --batch 16 2>/dev/null | tail -n 1; done
ROUNDS=2 BENCH_ARGS="--batch 1" bash tools/ab_env.sh "COFI_GEMM_INLAUNCH=0" "COFI_GEMM_INLAUNCH=1"
ROUNDS=1 bash tools/ab_env.sh "COFI_GEMM_INLAUNCH=0" "COFI_GEMM_INLAUNCH=1"
