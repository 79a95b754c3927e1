cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-kernel-timing --no-batch-sweep --batch 16 --steps 12 --warmup 4"
for e in X=1 COFI_KPCONV_AGG_PLANES=0 X=1 COFI_KPCONV_AGG_PLANES=0; do echo "== $e"; env $e $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'f/s')"; done
