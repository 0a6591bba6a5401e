#!/bin/bash
# the rolling batch of the 2-D forms in two lanes (host_inflight = -1) against the chunk scheme
cd "$GRAFT_REPO_ROOT" || exit 1
out=$PWD/gpurun_out/host_trace; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_large.py -q -x -k "rolling" 2>&1 | tail -3
for c in "c4 --members 8 --sweeps 500" "c4 --members 8 --sweeps 2000" "c4 --members 16 --sweeps 500" "c4 --members 32 --sweeps 500" "c4 --members 6 --sweeps 500" "c4 --members 64 --sweeps 500" "c4 --members 3 --sweeps 500"; do
  python tools/bench_host_pipeline.py $c --chunks 0 --inflight 0,-1 --reps 4 2>/dev/null | grep '^{' | cut -c1-60,94-200 | sed "s/^/$c | /"
done | tee $out/roll2d_lanes.txt
