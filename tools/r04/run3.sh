#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r04_run3
timeout 1200 python -m pytest tests/test_gpu_seam.py -q -x 2>&1 | tail -15 | tee gpurun_out/r04_run3/seam.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_small.py tests/test_gpu_scalar_cache.py -q -x 2>&1 | tail -5 | tee gpurun_out/r04_run3/parity.txt
