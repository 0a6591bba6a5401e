#!/bin/bash
# round 4: scheduler strategies on k_pipe2d (C2, C4) and k_fused2d<Gen> (C3)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$PWD
out=$R/gpurun_out/r04_run11
mkdir -p $out
for so in "" build/libxinv_ps_max-ilp.so build/libxinv_ps_iterative-minreg.so build/libxinv_ps_max-memory-clause.so build/libxinv_ps_iterative-maxocc.so; do
  tag=$(basename "${so:-main}" .so)
  export XINV_SO=${so:+$R/$so}
  [ -z "$so" ] && unset XINV_SO
  ( python tools/bench_configs.py c2 c3 c4 --reps 4 --sweeps 400; python tools/bench_configs.py c2 c4 --members 8 --reps 3 --sweeps 200 ) 2>/dev/null | grep '^{' | python -c "
import json,sys
print('$tag', ' | '.join('%s x%d %.4g (%.1f us)' % (json.loads(l)['config'], json.loads(l)['shape'][0], json.loads(l)['point_sweeps_per_s'], json.loads(l)['avg_launch_ms']*1e3) for l in sys.stdin))"
done 2>&1 | tee $out/summary.txt
