#!/bin/bash
# GPU box: quick A/B runs of the bench under environment switches.  usage: tools/ab_round.sh "NAME=VAL ..." ...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-kernel-timing --no-batch-sweep"
run() { echo "== $*"; env "$@" $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'f/s')"; }
run X=1
for cfg in "$@"; do run $cfg; done
run X=1
