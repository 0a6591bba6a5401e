#!/bin/bash
# the full GPU suite N times in a row on one box (fresh process each): flakiness check of the final state
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
N=${1:-10}
: > gpurun_out/r03_suite_repeats.txt
for i in $(seq 1 $N); do
  r=$(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -1)
  echo "run $i: $r" | tee -a gpurun_out/r03_suite_repeats.txt
done
