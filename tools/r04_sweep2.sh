cd /tmp; export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-kernel-timing --no-batch-sweep --no-f32 --no-steady"
for cfg in "16 4" "16 3" "16 6" "16 8" "32 2" "32 4" "8 4" "8 8" "64 2" "16 4"; do
  set -- $cfg
  steps=$(( 640 / $1 ))
  $B --batch $1 --inflight $2 --steps $steps --distinct-frames 32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$1 S=$2', round(d['value'],1), 'f/s', d['peak_mem_GB'], 'GB')"
done
