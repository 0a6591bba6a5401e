#!/bin/bash
# round 4: the ring variant of k_pipe3d as shipped -- suite, rates at 2 / 15 volumes and the 601x300x300 shape, traffic
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$PWD
out=$R/gpurun_out/r04_run10
mkdir -p $out
( time timeout 1800 python -m pytest tests -m gpu -q -x ) > $out/gputests.txt 2>&1
tail -4 $out/gputests.txt
( python tools/bench_configs.py c5 ofes --reps 3; python tools/bench_configs.py c5 --members 15 --reps 3; python tools/bench_configs.py c5 --members 4 --reps 3 ) 2>/dev/null | grep '^{' | tee $out/configs_3d.txt | cut -c1-200
db() { find "$1" -name '*.db' | head -1; }
cmd="python $R/tools/bench_configs.py c5 --members 15 --reps 1"
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/q_kt -o r -- $cmd > /dev/null 2>&1
python $R/tools/prof_summary.py kernels $(db /tmp/q_kt) $out/r04_kernel_trace_c5.txt | head -3 | cut -c1-150
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d /tmp/q_s -o r -- $cmd > /dev/null 2>&1
python $R/tools/prof_summary.py counters $(db /tmp/q_s) $out/r04_pmc_sq_issue_c5.txt | grep "k_pipe3d" | cut -c1-30,60-130
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/q_f -o r -- $cmd > /dev/null 2>&1
python $R/tools/prof_summary.py counters $(db /tmp/q_f) $out/r04_pmc_fetch_c5.txt | grep "k_pipe3d" | cut -c1-30,60-130
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/q_w -o r -- $cmd > /dev/null 2>&1
python $R/tools/prof_summary.py counters $(db /tmp/q_w) $out/r04_pmc_write_c5.txt | grep "k_pipe3d" | cut -c1-30,60-130
python $R/tools/prof_summary.py traffic $(db /tmp/q_f) $(db /tmp/q_w) "k_pipe3d" std3d_pipe3d_c5x15 $out/traffic_c5.json | head -2
