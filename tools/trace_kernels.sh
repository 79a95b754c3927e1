#!/bin/bash
# GPU box: kernel trace of a short bench run (one frame stream), by-grid table filtered by a pattern.  usage: tools/trace_kernels.sh <pattern> [ENV=VAL ...]
PAT=${1:-kpconv}; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o x -- python $R/bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-batch-sweep --inflight 1 > /tmp/prof_t.log 2>&1
DB=$(find /tmp/prof_t -name '*_results.db' | head -1)
python $R/tools/rocpd_summary.py $DB --by-grid | grep -E "$PAT|total kernel" | cut -c1-230
rm -rf /tmp/prof_t
