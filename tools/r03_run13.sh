#!/bin/bash
# round 3: full GPU suite after the planner change (pipelined pass at every size) and the removal of k_fused3d2 / spilling variants
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 2700 python -m pytest tests -m gpu -q -x ) > gpurun_out/r03_gputests_13.txt 2>&1
tail -6 gpurun_out/r03_gputests_13.txt
python tools/bench_configs.py c2 --members 8 --reps 2 2>/dev/null | grep '^{' | cut -c1-250
