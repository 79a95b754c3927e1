#!/bin/bash
# GPU box experiment: TWO bench processes sharing the GPU (each its own HIP hardware queues); sum of their rates vs one process.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-kernel-timing --no-batch-sweep --steps 600 --warmup 30"
one() { $B "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1))"; }
echo "one process, 4 streams: $(one)"
for infl in 2 3 4; do
  one --inflight $infl > /tmp/a.txt & one --inflight $infl > /tmp/b.txt & wait
  echo "two processes x $infl streams: $(cat /tmp/a.txt) + $(cat /tmp/b.txt)"
done
