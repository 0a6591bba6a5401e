#!/bin/bash
# round 3: where do C3 (Stommel, Munk) and the small-slice batches spend their time?  kernel trace + SQ issue counters
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/prof_r03b
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name '*.db' | head -1; }
run() {
  name=$1; shift
  rocprofv3 --kernel-trace --stats -d /tmp/q_kt_$name -o r -- "$@" > $out/${name}_bench.log 2>&1
  python $R/tools/prof_summary.py kernels $(db /tmp/q_kt_$name) $out/r03_kernel_trace_$name.txt | head -4 | cut -c1-200
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d /tmp/q_s_$name -o r -- "$@" > /dev/null 2>&1
  python $R/tools/prof_summary.py counters $(db /tmp/q_s_$name) $out/r03_pmc_sq_issue_$name.txt | grep "k_fused\|k_pipe" | cut -c1-30,60-130
  rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM -d /tmp/q_t_$name -o r -- "$@" > /dev/null 2>&1
  python $R/tools/prof_summary.py counters $(db /tmp/q_t_$name) $out/r03_pmc_sq_mem_$name.txt | grep "k_fused\|k_pipe" | cut -c1-30,60-130
}
run c3 python $R/tools/bench_configs.py c3 c3m --reps 1
run small python $R/tools/bench_small_batch.py
