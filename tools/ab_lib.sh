#!/bin/bash
# GPU box: A/B of two BUILDS of libcofi_hip.so on one box (compile-time changes): the named libraries (paths relative to cofii2p_amd/) are
# copied over libcofi_hip.so in turn, alternated `ROUNDS` times (CMD="python tools/x.py" runs that instead of the bench loop).
# usage: ROUNDS=2 tools/ab_lib.sh libcofi_hip_prev.so libcofi_hip_new.so
set -eu
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-kernel-timing --no-batch-sweep --no-f32 --no-steady --steps ${STEPS:-40} --warmup 5 ${BENCH_ARGS:-}"
cp cofii2p_amd/libcofi_hip.so /tmp/libcofi_hip_keep.so
trap 'cp /tmp/libcofi_hip_keep.so cofii2p_amd/libcofi_hip.so' EXIT   # an interrupted run must not leave a variant installed as the product library
for r in $(seq 1 ${ROUNDS:-2}); do
  for l in "$@"; do
    cp cofii2p_amd/$l cofii2p_amd/libcofi_hip.so
    echo -n "round $r  [$l]  "
    if [ -n "${CMD:-}" ]; then ($CMD 2>/dev/null || true) | tail -${CMD_LINES:-3}; continue; fi
    ($B 2>/dev/null || true) | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'f/s')"
  done
done
