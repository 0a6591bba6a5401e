#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06_repeat
for i in 1 2; do ( timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -2 ) | tee -a gpurun_out/r06_repeat/suite_repeats.txt; done
( timeout 1500 python tests/fuzz_lanes.py 40000 4000 2>&1 | tail -3 ) | tee gpurun_out/r06_repeat/fuzz_lanes_4000.txt
( timeout 600 python tests/fuzz_plan.py 2>&1 | tail -2 ) | tee -a gpurun_out/r06_repeat/fuzz_lanes_4000.txt
