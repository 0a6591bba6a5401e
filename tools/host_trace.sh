#!/bin/bash
# Where a host-pointer call's wall clock goes: the library's own phase stamps (XINV_HOST_TRACE=1, test-hooks build) for the
# C2 front-end call and the C4 x 8 / C5 x 15 host-pointer solves, then the shipped library's timings of the same calls.
#   gpurun --timeout 900 -- 'bash tools/host_trace.sh'      -> gpurun_out/host_trace/*
cd "$GRAFT_REPO_ROOT" || exit 1
out=$PWD/gpurun_out/host_trace; mkdir -p $out
H=$PWD/build/libxinv_hooks.so
for cfg in "c4 --members 8 --sweeps 500 --chunks 0,8,4" "c5 --members 15 --sweeps 200 --chunks 0"; do
  set -- $cfg
  XINV_SO=$H XINV_HOST_TRACE=1 python tools/bench_host_pipeline.py $cfg --reps 2 > $out/trace_$1.txt 2>&1
  python tools/bench_host_pipeline.py $cfg --reps 5 2>/dev/null | grep '^{' > $out/time_$1.txt
  cut -c1-220 $out/time_$1.txt
done
XINV_SO=$H XINV_HOST_TRACE=1 python tools/c2_e2e.py 2 > $out/trace_c2.txt 2>&1
python tools/c2_e2e.py 9 2>/dev/null | tee $out/time_c2.txt
tail -14 $out/trace_c2.txt | cut -c1-160
tail -40 $out/trace_c4.txt | cut -c1-160
