cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-kernel-timing --no-batch-sweep"
run() { echo "== $*"; env $1 $B ${@:2} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'f/s')"; }
run X=1
run X=1 --steps 20 --warmup 5
run COFI_ASYNC_BRANCH_MASK=1 --inflight 2
run COFI_ASYNC_BRANCH_MASK=1 --inflight 2 --steps 20 --warmup 5
run COFI_ASYNC_BRANCH_MASK=1 --inflight 4
run COFI_ASYNC_BRANCH_MASK=1 --inflight 3
run COFI_ASYNC_BRANCH_MASK=7 --inflight 2
run X=1 --inflight 4 --slots-per-stream 1 --steps 20 --warmup 5
run X=1 --inflight 4 --slots-per-stream 3 --steps 20 --warmup 5
