#!/bin/bash
# round 6: k_pipe3d tail cut -- parity of the mixed launch, then C5 rates at 13..16 volumes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=$PWD/gpurun_out/r06_tail
mkdir -p $out
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "3d or tail" 2>&1 | tail -5 ) | tee $out/tests.txt
( timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_plan.py tests/test_gpu_seam.py -q -x -k "3d or omega or c5 or C5 or std3d" 2>&1 | tail -3 ) | tee -a $out/tests.txt
for m in 13 14 15 16 17 30; do python tools/bench_configs.py c5 --members $m --reps 2 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['shape'][0], '%.4g' % d['point_sweeps_per_s'], 'launch %.1f us' % (d['avg_launch_ms'] * 1e3), {k: d.get(k) for k in ('k_chunks', 'cut_tiles')})"; done | tee $out/c5_members.txt
