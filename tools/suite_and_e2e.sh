#!/bin/bash
# GPU suite + the host-to-host timings (between full tools/final.sh runs)
cd "$GRAFT_REPO_ROOT" || exit 1
out=$PWD/gpurun_out/suite; mkdir -p $out
( time timeout 2400 python -m pytest tests -m gpu -q -x ) > $out/gputests.txt 2>&1; tail -5 $out/gputests.txt
python tools/c2_e2e.py 9 | tee $out/c2_e2e.txt
python tools/bench_host_pipeline.py c4 --members 8 --sweeps 500 --chunks 0 --reps 5 2>/dev/null | grep '^{' | cut -c1-220 | tee $out/c4x8.txt
python tools/bench_host_pipeline.py c5 --members 15 --sweeps 200 --chunks 0 --reps 3 2>/dev/null | grep '^{' | cut -c1-220 | tee $out/c5x15.txt
