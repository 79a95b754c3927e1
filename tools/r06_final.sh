#!/bin/bash
# GPU box (round 6): tests of the touched paths, the default bench, then traces + PMC passes (tools/profile_r06.sh)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_forward_gpu.py -q -x -k "f16x3 or big_tiles or stress or kitti" 2>&1 | tail -5 > gpurun_out/r06/t_final_quick.txt
tail -5 gpurun_out/r06/t_final_quick.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench_d.json 2> gpurun_out/r06/bench_d.err
python -c "
import json
d=json.loads(open('gpurun_out/r06/bench_d.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d.get('roofline_f16x3_kernel'))
print(d['config'].get('batch1_frames_per_s'), d['forward_sync']['frames_per_s'], d['stress_config']['frames_per_s'])
"
tools/profile_r06.sh b2258e2 > gpurun_out/r06/profile.log 2>&1
tail -20 gpurun_out/r06/profile.log
