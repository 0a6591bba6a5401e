#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r04_run6
rm -f gpurun_out/r04_run6/stop_rule.txt
for lag in 1 0; do for cfg in c2 c5; do
  XINV_LAG=$lag timeout 900 python tests/stop_rule_edge.py $cfg gpurun_out/r04_run6/stop_rule.txt 2>&1 | tail -22
done; done
