#!/bin/bash
# The sweep loop in lanes (profiles/r04_lanes.txt): 1 against 2 launch chains over batch sizes of every
# configuration, on one box.  XINV_LANES=n forces n lanes (0 / 1: off; unset: the engine's rule, lane_rule in xinv_hip.hip).
#   gpurun -- 'bash tools/r04/lanes.sh'            -> gpurun_out/lanes/out.txt
cd "${GRAFT_REPO_ROOT:-$PWD}" || exit 1
mkdir -p gpurun_out/lanes
run() { python tools/bench_configs.py "$@" --reps 3 2>/dev/null | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$TAG', d['config'], d['shape'], '%.4g  launch %.1f us' % (d['point_sweeps_per_s'], d['avg_launch_ms']*1e3))"; }
{
for cfg in "c2 --members 2" "c2 --members 3" "c2 --members 4" "c2 --members 6" "c2 --members 8" "c2 --members 16" "c2 --members 32" \
           "c4 --members 2" "c4 --members 3" "c4 --members 4" "c4 --members 8" "c4 --members 16" "c4 --members 64" "c4 --members 128" \
           "c1 --members 16" "c1 --members 64" "c1 --members 365" "c1 --members 1000" \
           "c5 --members 2" "c5 --members 4" "c5 --members 8" "c5 --members 15" "c5 --members 30"; do
  for n in 1 2; do TAG=lanes$n XINV_LANES=$n run $cfg; done
  TAG=rule run $cfg
done
XINV_LANES=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_small.py tests/test_gpu_large.py tests/test_gpu_watchdog.py tests/test_gpu_lanes.py -q -x 2>&1 | tail -2
} > gpurun_out/lanes/out.txt 2>&1
cat gpurun_out/lanes/out.txt
