#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > gpurun_out/r03_gputests_5.txt 2>&1
tail -6 gpurun_out/r03_gputests_5.txt
( python tools/bench_host_pipeline.py c5 --members 15 --sweeps 200; python tools/bench_host_pipeline.py c4 --members 8 --sweeps 500 ) > gpurun_out/r03_host_pipeline.txt 2>&1
cat gpurun_out/r03_host_pipeline.txt | cut -c1-260
