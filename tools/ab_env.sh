#!/bin/bash
# GPU box: A/B runs of the headline loop under different ENVIRONMENT settings (each quoted string = VAR=value pairs), alternated `ROUNDS` times
# in one call (same box).  usage: ROUNDS=2 tools/ab_env.sh "COFI_GEMM_BIG=0" "COFI_GEMM_BIG=1" ...
set -eu
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-kernel-timing --no-batch-sweep --no-f32 --no-steady --steps ${STEPS:-40} --warmup 5 ${BENCH_ARGS:-}"
for r in $(seq 1 ${ROUNDS:-2}); do
  for e in "$@"; do
    echo -n "round $r  [$e]  "
    (env $e $B 2>/dev/null || true) | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'f/s   batch1', round(d['config'].get('batch1_frames_per_s') or 0,1))"
  done
done
