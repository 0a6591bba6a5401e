#!/bin/bash
# staging memcpy workers (hooks build: XINV_COPY_THREADS) against the upload span
cd "$GRAFT_REPO_ROOT" || exit 1
out=$PWD/gpurun_out/host_trace; mkdir -p $out

export XINV_SO=$PWD/build/libxinv_hooks.so
for t in -1 1 2 3 4 -1 1 2 3 4; do
  export XINV_COPY_THREADS=$t
  python tools/c2_e2e.py 7 | sed "s/^/threads $t /"
  python tools/bench_host_pipeline.py c4 --members 8 --sweeps 500 --chunks 0 --reps 4 2>/dev/null | grep '^{' | grep -v resident | cut -c94-260 | sed "s/^/threads $t c4x8 /"
  python tools/bench_host_pipeline.py c5 --members 15 --sweeps 200 --chunks 0 --reps 2 2>/dev/null | grep "^{" | grep -v resident | cut -c94-260 | sed "s/^/threads $t c5x15 /"
done | tee $out/copy_threads.txt
