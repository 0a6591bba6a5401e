#!/bin/bash
# round 3, first GPU call: scalar-cache probe, whole GPU suite (with the new C1 / reuse-stress tests), default bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time ./build/scache_probe 1500 ) > gpurun_out/r03_scache_probe.txt 2>&1
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r03_gputests_1.txt 2>&1
tail -5 gpurun_out/r03_gputests_1.txt
( time python bench.py ) > gpurun_out/r03_bench_0.txt 2>&1
tail -c 1500 gpurun_out/r03_bench_0.txt
cat gpurun_out/r03_scache_probe.txt
