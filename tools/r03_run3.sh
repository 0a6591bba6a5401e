#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > gpurun_out/r03_gputests_3.txt 2>&1
tail -6 gpurun_out/r03_gputests_3.txt
( time python bench.py ) > gpurun_out/r03_bench_1.txt 2>&1
tail -c 600 gpurun_out/r03_bench_1.txt
