#!/bin/bash
# round 3: k_pipe2d with EXEC-masked update / norm, incremental row pointers: parity tests + A/B against the previous build
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q ) > gpurun_out/r03_gputests_10.txt 2>&1
tail -4 gpurun_out/r03_gputests_10.txt
for so in "" build/libxinv_noexec.so ""  build/libxinv_noexec.so; do XINV_SO=$so python tools/ab_pipe.py 0 30 36 60 c4 c1; done 2>&1 | grep -v amdgpu | tee gpurun_out/r03_execsel_ab2.txt
