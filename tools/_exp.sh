cd $GRAFT_REPO_ROOT
export COFI_GEMM=bf16x6
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "big_tiles" 2>&1 | tail -3
PROBE_MINN=33 PROBE_MAXN=64 PROBE_MINK=96 PROBE_KS=1,2,3 python tools/big_gemm_probe.py > gpurun_out/r05/big_probe_n64.txt 2>/dev/null
cat gpurun_out/r05/big_probe_n64.txt
