#!/bin/bash
# GPU box: A/B runs of the bench under different ARGUMENT sets (each quoted string = extra bench.py arguments).  usage: tools/ab_args.sh "" "--copy-inputs" ...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-kernel-timing --no-batch-sweep"
for a in "$@"; do echo "== $a"; $B $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'f/s')"; done
