cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for e in X=1 COFI_GEMM_KW=1:0; do echo "== $e"; env $e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'f/s; forward_sync', round(d['forward_sync']['frames_per_s'],1), '; batch16', round(d['stack_mode_batches']['batch_16']['frames_per_s'],1), '; batch4', round(d['stack_mode_batches']['batch_4']['frames_per_s'],1), '; stress ms', round(d['stress_config']['ms_per_frame'],2))"; done
