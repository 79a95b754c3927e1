#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-gaps}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-timing --no-batch-sweep"
for cfg in "if4:--inflight 4" "if1:--inflight 1" "if2:--inflight 2"; do
  name=${cfg%%:*}; extra=${cfg#*:}
  rocprofv3 --kernel-trace -d /tmp/prof_$name -o x -- $BENCH $extra > $OUT/$name.log 2>&1
  DB=$(find /tmp/prof_$name -name '*_results.db' | head -1)
  echo "=== $name"; grep -o '"value": [0-9.]*' $OUT/$name.log | head -1
  python $R/tools/rocpd_gaps.py $DB | tee $OUT/${name}_gaps.md
  rm -rf /tmp/prof_$name
done
