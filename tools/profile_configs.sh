#!/bin/bash
# Kernel trace + HBM-side traffic (separate FETCH_SIZE / WRITE_SIZE passes) of the other BASELINE
# configurations and the 9-point forms:  gpurun -- 'bash tools/profile_configs.sh r01'
tag=${1:-r01}
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name '*.db' | head -1; }
run() {   # name, command...
  name=$1; shift
  rocprofv3 --kernel-trace --stats -d /tmp/q_kt_$name -o r -- "$@" > $out/${name}_bench.log 2>&1
  python $R/tools/prof_summary.py kernels $(db /tmp/q_kt_$name) $out/${tag}_kernel_trace_$name.txt > /dev/null
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/q_f_$name -o r -- "$@" > /dev/null 2>&1
  python $R/tools/prof_summary.py counters $(db /tmp/q_f_$name) $out/${tag}_pmc_fetch_$name.txt > /dev/null
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/q_w_$name -o r -- "$@" > /dev/null 2>&1
  python $R/tools/prof_summary.py counters $(db /tmp/q_w_$name) $out/${tag}_pmc_write_$name.txt > /dev/null
  grep '^{' $out/${name}_bench.log | cut -c1-330
}
run configs python $R/tools/bench_configs.py c1 c3 c3m c4 c5 c5g --reps 1
run ofes python $R/tools/bench_configs.py ofes --reps 1
run ninepoint python $R/tools/bench_ninepoint.py
