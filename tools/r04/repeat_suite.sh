#!/bin/bash
# round 4: the GPU suite several times over (fresh processes), and the 3-D / seam / contracted tests many times: intermittent failures?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=gpurun_out/r04_repeat
mkdir -p $out
: > $out/r04_suite_repeats.txt
for i in 1 2 3; do
  timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2 | tee -a $out/r04_suite_repeats.txt
done
for i in $(seq 1 12); do
  timeout 600 python -m pytest tests/test_gpu_small.py tests/test_gpu_seam.py tests/test_gpu_fma.py -q -x -p no:cacheprovider 2>&1 | tail -1 | tee -a $out/r04_suite_repeats.txt
  timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -x -p no:cacheprovider -k "omega or c5 or 3d" 2>&1 | tail -1 | tee -a $out/r04_suite_repeats.txt
  # (the sweep loop in lanes: every lane count, members stopping apart; batches of the large-grid tests)
  timeout 600 python -m pytest tests/test_gpu_lanes.py tests/test_gpu_large.py -q -x -p no:cacheprovider 2>&1 | tail -1 | tee -a $out/r04_suite_repeats.txt
done
