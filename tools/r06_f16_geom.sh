#!/bin/bash
# GPU box (round 6): the f16x3 kernel in its eight-wave (default) and four-wave (cofi_tune_big_debug 256) geometry: tests, then the per-shape probe
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "f16x3 or big_tiles" 2>&1 | tail -30 > gpurun_out/r06/t_f16_geom.txt
tail -5 gpurun_out/r06/t_f16_geom.txt
PROBE_DBG=0 timeout 600 python tools/f16_probe.py > gpurun_out/r06/f16_probe_w8.txt 2>&1
PROBE_DBG=256 timeout 600 python tools/f16_probe.py > gpurun_out/r06/f16_probe_w4.txt 2>&1
grep -v "SAME BITS" gpurun_out/r06/f16_probe_w8.txt | tail -22
grep -v "SAME BITS" gpurun_out/r06/f16_probe_w4.txt | tail -22
