#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-kernel-timing --no-batch-sweep"
run() { echo "== q=$1 streams=$2 $3 $4 $5 $6"; env GPU_MAX_HW_QUEUES=$1 COFI_BENCH_STREAMS=$2 $B $3 $4 $5 $6 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'f/s')"; }
run 4 capnull --inflight 4
run 5 capnull --inflight 5
run 6 capnull --inflight 6
run 8 capnull --inflight 6
run 8 capnull --inflight 8
run 8 capnull --inflight 5
run 4 capnull --inflight 5
run 4 capnull --inflight 4 --steps 20 --warmup 5
run 4 capnull --inflight 4 --steps 20 --warmup 5
run 4 capnull --inflight 4 --steps 40 --warmup 5
