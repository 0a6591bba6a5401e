#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
( timeout 2400 python -m pytest tests/test_gpu_large.py -q -x -k "rolling" 2>&1 | tail -12 )
python tools/bench_host_pipeline.py c4 --members 8 --sweeps 500 --chunks 0,2 --inflight 0 2>/dev/null | grep "^{" | cut -c1-170
python tools/bench_host_pipeline.py c5 --members 15 --sweeps 200 --chunks 0 --inflight 0 2>/dev/null | grep "^{" | cut -c1-170
