#!/bin/bash
# GPU box: alternate the bench of two source trees (the repo and its _ab/ worktree of an earlier commit) on ONE box.  usage: tools/ab_dirs.sh "<bench args>" [rounds]
R=$GRAFT_REPO_ROOT; ARGS=${1:-}; N=${2:-2}
cd /tmp; export TMPDIR=/tmp
for i in $(seq $N); do
  for d in _ab .; do
    python $R/$d/bench.py --no-cpu-baseline --no-kernel-timing --no-batch-sweep --no-f32 --no-steady $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$d', '$ARGS', round(d['value'],1))"
  done
done
