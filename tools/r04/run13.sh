#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$PWD
out=$R/gpurun_out/r04_run13
mkdir -p $out
db() { find "$1" -name '*.db' | head -1; }
cmd="python $R/tools/bench_configs.py c5 --members 15 --reps 1"
for so in "" build/libxinv_p3touch.so; do
  tag=$(basename "${so:-main}" .so)
  export XINV_SO=${so:+$R/$so}
  [ -z "$so" ] && unset XINV_SO
  python tools/bench_configs.py c5 --members 15 --reps 3 2>/dev/null | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$tag', 'c5x15 %.4g  launch %.1f us' % (d['point_sweeps_per_s'], d['avg_launch_ms']*1e3))"
  timeout 600 python -m pytest tests/test_gpu_small.py -q -x -k "two_sweeps" 2>&1 | tail -1
  ( cd /tmp; rm -rf /tmp/q_f /tmp/q_w
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/q_f -o r -- $cmd > /dev/null 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/q_w -o r -- $cmd > /dev/null 2>&1
    python $R/tools/prof_summary.py traffic $(db /tmp/q_f) $(db /tmp/q_w) "k_pipe3d" std3d_pipe3d_$tag $out/traffic_$tag.json | head -1 )
done 2>&1 | tee $out/summary.txt
