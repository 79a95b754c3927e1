#!/bin/bash
# GPU box: frames/s of stack-mode micro-batches (B frames per submission, S submissions in flight) in one arithmetic + kernel rooflines per B
set -u
R=$GRAFT_REPO_ROOT
M=${1:-bf16x6}
OUT=$R/gpurun_out/r04_sweep_$M
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "1 4" "2 4" "4 2" "4 4" "8 2" "8 4" "16 2" "16 4"; do
  set -- $cfg; B=$1; S=$2
  steps=$(( 256 / B )); [ $steps -lt 12 ] && steps=12
  extra="--no-kernel-timing"; [ "$S" = "2" ] && extra=""
  python $R/bench.py --no-cpu-baseline --no-batch-sweep --no-f32 $extra --gemm $M --batch $B --inflight $S --steps $steps > $OUT/b${B}_s${S}.json 2> $OUT/b${B}_s${S}.err
  python - <<PY
import json
d = json.loads(open("$OUT/b${B}_s${S}.json").read().strip().splitlines()[-1])
ra, rc, rf = d.get("roofline_attention", {}), d.get("roofline_cross_attention", {}), d.get("roofline", {})
print("B=$B S=$S  %.1f f/s  peak_mem %.1f GB  gemm frac %s  attn %s  cross %s" % (d["value"], d.get("peak_mem_GB", 0), rf.get("frac"), ra.get("frac"), rc.get("frac")))
PY
done
