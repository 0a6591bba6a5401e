#!/bin/bash
# Profiles of the bench.py workload on the GPU box: kernel trace, HBM-side traffic (separate
# FETCH_SIZE / WRITE_SIZE passes, as MI355X_MICROARCH.md prescribes) and an SQ issue pass.
#   gpurun -- 'bash tools/profile_headline.sh r01'      -> gpurun_out/prof_<tag>/*.txt, traffic.json
tag=${1:-r03}
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cmd="python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-parity --no-configs"
db() { find "$1" -name '*.db' | head -1; }
rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o r -- $cmd > $out/bench_kernel_trace.log 2>&1
python $R/tools/prof_summary.py kernels $(db /tmp/p_kt) $out/${tag}_kernel_trace_c2.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_f -o r -- $cmd > /dev/null 2>&1
python $R/tools/prof_summary.py counters $(db /tmp/p_f) $out/${tag}_pmc_fetch_c2.txt
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_w -o r -- $cmd > /dev/null 2>&1
python $R/tools/prof_summary.py counters $(db /tmp/p_w) $out/${tag}_pmc_write_c2.txt
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d /tmp/p_s -o r -- $cmd > /dev/null 2>&1
python $R/tools/prof_summary.py counters $(db /tmp/p_s) $out/${tag}_pmc_sq_issue_c2.txt
cp $R/profiles/traffic.json $out/traffic.json 2>/dev/null
python $R/tools/prof_summary.py traffic $(db /tmp/p_f) $(db /tmp/p_w) "k_pipe2d<FusedStd2D, 3u, false, 1" std2d_pipe_um3 $out/traffic.json
XINV_PIPE=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_f0 -o r -- $cmd --no-hbm > /dev/null 2>&1
XINV_PIPE=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_w0 -o r -- $cmd --no-hbm > /dev/null 2>&1
python $R/tools/prof_summary.py traffic $(db /tmp/p_f0) $(db /tmp/p_w0) "k_fused2d<FusedStd2D, 4" std2d_spl4_um3 $out/traffic.json
# the HBM-bound variant of the same run (bench.py roofline_hbm): one sweep per pass, all arrays streamed, every tile
python $R/tools/prof_summary.py traffic $(db /tmp/p_f) $(db /tmp/p_w) "k_fused2d<FusedStd2D, 1" std2d_spl1_um0_all $out/traffic.json
# C4 (BASELINE configs[3]): the general-form pipelined pass on 64 members (1.6 GB working set: real HBM traffic)
c4="python $R/bench.py --config c4 --members 64 --steps 2 --warmup 1 --sweeps 100"
rocprofv3 --kernel-trace --stats -d /tmp/p_kt4 -o r -- $c4 > $out/bench_kernel_trace_c4.log 2>&1
python $R/tools/prof_summary.py kernels $(db /tmp/p_kt4) $out/${tag}_kernel_trace_c4.txt > /dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_f4 -o r -- $c4 > /dev/null 2>&1
python $R/tools/prof_summary.py counters $(db /tmp/p_f4) $out/${tag}_pmc_fetch_c4.txt > /dev/null
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_w4 -o r -- $c4 > /dev/null 2>&1
python $R/tools/prof_summary.py counters $(db /tmp/p_w4) $out/${tag}_pmc_write_c4.txt > /dev/null
python $R/tools/prof_summary.py traffic $(db /tmp/p_f4) $(db /tmp/p_w4) "k_pipe2d<FusedGen2D, 31u, true, 1" gen2d_pipe_um31_fr_c4x64 $out/traffic.json
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d /tmp/p_s4 -o r -- $c4 > /dev/null 2>&1
python $R/tools/prof_summary.py counters $(db /tmp/p_s4) $out/${tag}_pmc_sq_issue_c4.txt > /dev/null
# the HBM leg of bench.py (8 members with their own A, C, F: 2 GB): its kernel has grid_y = 8
head -4 $out/${tag}_kernel_trace_c2.txt | cut -c1-160
head -3 $out/${tag}_kernel_trace_c4.txt | cut -c1-160
grep "k_fused2d\|k_pipe2d" $out/${tag}_pmc_sq_issue_c2.txt | cut -c1-30,60-140
cat $out/traffic.json
