#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r04_run14
( time timeout 2400 python -m pytest tests/test_gpu_watchdog.py tests/test_gpu_stop_rule.py -q -x ) 2>&1 | tail -25 | tee gpurun_out/r04_run14/watchdog.txt
XINV_SO=$PWD/build/libxinv_hooks.so XINV_HOOKS_SUITE=1 timeout 1200 python -m pytest tests/hooks_suite -q -x -m gpu 2>&1 | tail -8
