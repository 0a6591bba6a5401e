#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06_fuzz
( time timeout 2400 python tests/fuzz_lanes.py 10000 3000 2>&1 | tail -12 ) 2>&1 | tee gpurun_out/r06_fuzz/fuzz_lanes_3000.txt
( time timeout 900 python tests/fuzz_lanes.py 20000 600 --single 2>&1 | tail -6 ) 2>&1 | tee -a gpurun_out/r06_fuzz/fuzz_lanes_3000.txt
