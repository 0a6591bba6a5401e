#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$PWD
out=$R/gpurun_out/r04_run9
mkdir -p $out
for so in "" build/libxinv_p3frm.so build/libxinv_p3frmd1.so build/libxinv_p3frmd2.so; do
 for unal in 0 1; do
  tag=$(basename "${so:-main}" .so)_unal$unal
  export XINV_SO=${so:+$R/$so}
  [ -z "$so" ] && unset XINV_SO
  XINV_P3_UNAL=$unal python tools/bench_configs.py c5 --members 15 --reps 3 2>/dev/null | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$tag', 'c5x15 %.4g  launch %.1f us' % (d['point_sweeps_per_s'], d['avg_launch_ms']*1e3))"
  XINV_P3_UNAL=$unal timeout 600 python -m pytest tests/test_gpu_small.py -q -x -k "two_sweeps" 2>&1 | tail -1
 done
done 2>&1 | tee $out/summary.txt
