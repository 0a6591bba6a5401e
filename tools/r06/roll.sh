#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
( timeout 2400 python -m pytest tests/test_gpu_large.py tests/test_gpu_frontend.py tests/test_gpu_multidev.py tests/test_gpu_f32.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -6 )
for i in 1 2; do python tools/bench_host_pipeline.py c5 --members 15 --sweeps 200 --chunks 0 --inflight 0 2>/dev/null | grep "^{" | cut -c1-170; done
