#!/bin/bash
# round 4: k_pipe2d with a workgroup barrier every FOUR steps (ring of eight rows) against every two
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$PWD
out=$R/gpurun_out/r04_run12
mkdir -p $out
for so in "" build/libxinv_pb4.so; do
  tag=$(basename "${so:-main}" .so)
  export XINV_SO=${so:+$R/$so}
  [ -z "$so" ] && unset XINV_SO
  ( python tools/bench_configs.py c2 c4 --reps 4 --sweeps 400; python tools/bench_configs.py c2 --members 8 --reps 3 --sweeps 200; python tools/bench_configs.py c4 --members 64 --reps 3 --sweeps 100; python tools/bench_configs.py c1 --members 64 --reps 3 ) 2>/dev/null | grep '^{' | python -c "
import json,sys
print('$tag', ' | '.join('%s x%d %.4g (%.1f us)' % (json.loads(l)['config'], json.loads(l)['shape'][0], json.loads(l)['point_sweeps_per_s'], json.loads(l)['avg_launch_ms']*1e3) for l in sys.stdin))"
  timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "pipelined" 2>&1 | tail -1
  timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k "c2 or c4 or poisson or gill" 2>&1 | tail -1
done 2>&1 | tee $out/summary.txt
