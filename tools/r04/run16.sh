#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/r04_run16
for so in "" build/libxinv_gwu.so "" build/libxinv_gwu.so; do
  tag=$(basename "${so:-main}" .so)
  export XINV_SO=${so:+$R/$so}
  [ -z "$so" ] && unset XINV_SO
  python tools/bench_configs.py c3 --reps 5 --sweeps 300 2>/dev/null | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$tag', 'c3 %.4g  launch %.2f us rows %d' % (d['point_sweeps_per_s'], d['avg_launch_ms']*1e3, d['rows_per_tile']))"
done 2>&1 | tee gpurun_out/r04_run16/summary.txt
XINV_SO=$R/build/libxinv_gwu.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "gen2d" 2>&1 | tail -1
