#!/bin/bash
# GPU box: the driver's bench command, the default one, and kernel traces.  usage: tools/bench_round.sh <tag> [trace]
set -u
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver.json 2> $OUT/bench_driver.err
python $R/bench.py --no-cpu-baseline --no-kernel-timing --no-batch-sweep > $OUT/bench_200.json 2> $OUT/bench_200.err
if [ "${2:-}" = "trace" ]; then
  BENCH="python $R/bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-batch-sweep"
  for cfg in "inflight3:" "inflight1:--inflight 1"; do
    name=${cfg%%:*}; extra=${cfg#*:}
    rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o x -- $BENCH $extra > $OUT/$name.log 2>&1
    DB=$(find /tmp/prof_$name -name '*_results.db' | head -1)
    python $R/tools/rocpd_summary.py $DB > $OUT/${name}_kernel_trace.md 2>&1
    python $R/tools/rocpd_summary.py $DB --by-grid > $OUT/${name}_by_grid.md 2>&1
    rm -rf /tmp/prof_$name
  done
fi
python - <<PY
import json
for f in ("bench_driver", "bench_200"):
    try:
        d = json.loads(open("$OUT/%s.json" % f).read().strip().splitlines()[-1])
        print(f, round(d["value"], 1), "f/s", d.get("kernel_ms_per_frame"), d.get("roofline_attention"), d.get("stack_mode_batches"))
    except Exception as e:
        print(f, "failed", e); print(open("$OUT/%s.err" % f).read()[-2000:])
PY
tail -3 $OUT/inflight1_kernel_trace.md 2>/dev/null
