#!/bin/bash
# One GPU-box call: parity tests, then a short bench line.  Usage (from the repo root): gpurun -- 'bash tools/gpu_check.sh [tag] [pytest args]'
tag=${1:-run}; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q "$@" > gpurun_out/${tag}_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${tag}_tests.log
tail -25 gpurun_out/${tag}_tests.log
