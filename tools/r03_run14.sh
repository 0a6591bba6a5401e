#!/bin/bash
# round 3: SQ counters of the reworked biharmonic one-pass kernel, tile heights
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/prof_r03c
mkdir -p $out
cd $R; bash tools/ab_rows.sh c3m 9 21 24
cd /tmp && export TMPDIR=/tmp
db() { find "$1" -name '*.db' | head -1; }
cmd="python $R/tools/bench_configs.py c3m --reps 1"
rocprofv3 --kernel-trace --stats -d /tmp/q_kt -o r -- $cmd > /dev/null 2>&1
python $R/tools/prof_summary.py kernels $(db /tmp/q_kt) $out/r03_kernel_trace_c3m.txt | head -3 | cut -c1-150
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d /tmp/q_s -o r -- $cmd > /dev/null 2>&1
python $R/tools/prof_summary.py counters $(db /tmp/q_s) $out/r03_pmc_sq_issue_c3m.txt | grep "k_fusedbih" | cut -c1-30,60-130
rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS SQ_INST_LEVEL_VMEM -d /tmp/q_t -o r -- $cmd > /dev/null 2>&1
python $R/tools/prof_summary.py counters $(db /tmp/q_t) $out/r03_pmc_sq_more_c3m.txt | grep "k_fusedbih" | cut -c1-30,60-130
