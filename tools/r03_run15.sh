#!/bin/bash
# round 3: SQ counters and traffic of k_pipe3d (C5, 15 volumes)
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/prof_r03d
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name '*.db' | head -1; }
cmd="python $R/tools/bench_configs.py c5 --members 15 --reps 1"
rocprofv3 --kernel-trace --stats -d /tmp/q_kt -o r -- $cmd > /dev/null 2>&1
python $R/tools/prof_summary.py kernels $(db /tmp/q_kt) $out/r03_kernel_trace_c5.txt | head -3 | cut -c1-150
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d /tmp/q_s -o r -- $cmd > /dev/null 2>&1
python $R/tools/prof_summary.py counters $(db /tmp/q_s) $out/r03_pmc_sq_issue_c5.txt | grep "k_pipe3d" | cut -c1-30,60-130
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d /tmp/q_t -o r -- $cmd > /dev/null 2>&1
python $R/tools/prof_summary.py counters $(db /tmp/q_t) $out/r03_pmc_sq_more_c5.txt | grep "k_pipe3d" | cut -c1-30,60-130
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/q_f -o r -- $cmd > /dev/null 2>&1
python $R/tools/prof_summary.py counters $(db /tmp/q_f) $out/r03_pmc_fetch_c5.txt | grep "k_pipe3d" | cut -c1-30,60-130
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/q_w -o r -- $cmd > /dev/null 2>&1
python $R/tools/prof_summary.py counters $(db /tmp/q_w) $out/r03_pmc_write_c5.txt | grep "k_pipe3d" | cut -c1-30,60-130
