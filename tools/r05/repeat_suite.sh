#!/bin/bash
# round 5: the GPU suite twice more (fresh processes), the seam / small / plan / full-size-seam tests ten times, the ring
# kernels' alternation stress families five times: intermittent failures?   -> gpurun_out/r05_repeat/r05_suite_repeats.txt
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r05_repeat
mkdir -p $out
: > $out/r05_suite_repeats.txt
for i in 1 2; do
  timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -1 | tee -a $out/r05_suite_repeats.txt
done
for i in $(seq 1 10); do
  timeout 600 python -m pytest tests/test_gpu_small.py tests/test_gpu_seam.py tests/test_gpu_plan.py -q -x -p no:cacheprovider 2>&1 | tail -1 | tee -a $out/r05_suite_repeats.txt
  timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -x -p no:cacheprovider -k "seam or omega" 2>&1 | tail -1 | tee -a $out/r05_suite_repeats.txt
done
for i in $(seq 1 5); do
  for f in pipe2d_seam fused3d_seam_ring fused9_seam fused2d_seam fused3d_seam fused3dg_seam; do
    timeout 600 python tests/stress_scalar_cache.py $f 50 2>&1 | tail -1 | tee -a $out/r05_suite_repeats.txt
  done
done
