#!/bin/bash
# Sweeps of the host-pointer pipeline's knobs on one box (profiles/r06_host_pipeline.txt, sections (2)-(9), were made with
# variants of this loop): tools/bench_host_pipeline.py <config> --members M --sweeps S --chunks a,b,c --inflight i,j [--lanes n]
# [--check-every n]; the hooks build (XINV_SO=build/libxinv_hooks.so) additionally reads XINV_COPY_PRIO (0: plain copy streams),
# XINV_COPY_THREADS (staging memcpy workers; -1: none), XINV_INFLIGHT_2D (default chunk solves in flight of the 2-D forms) and
# XINV_HOST_TRACE (phase stamps on stderr).  As committed: chunk sizes with two chunk solves in flight.
# chunk sizes with TWO chunk solves in flight (the stable setting in bench.py's process)
cd "$GRAFT_REPO_ROOT" || exit 1
out=$PWD/gpurun_out/host_trace; mkdir -p $out
for c in "c4 --members 3 --sweeps 500 --chunks 1,3" "c4 --members 6 --sweeps 500 --chunks 1,2,3" "c4 --members 8 --sweeps 500 --chunks 1,2,4" "c4 --members 12 --sweeps 500 --chunks 2,3,4,6" "c4 --members 16 --sweeps 500 --chunks 2,4,8" "c4 --members 32 --sweeps 500 --chunks 2,4,8,16" "c4 --members 64 --sweeps 500 --chunks 8,16,32" "c2 --members 2 --sweeps 500 --chunks 1,2" "c2 --members 3 --sweeps 500 --chunks 1,3" "c2 --members 8 --sweeps 500 --chunks 1,2,4"; do
  python tools/bench_host_pipeline.py $c --inflight 2 --reps 3 2>/dev/null | grep '^{' | cut -c1-60,94-190 | sed "s/^/$c | /"
done | tee $out/chunks_2d_inflight2.txt
