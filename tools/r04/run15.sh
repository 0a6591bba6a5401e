#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/r04_run15
python tools/bench_configs.py c3 --reps 4 --sweeps 300 2>/dev/null | grep '^{' | cut -c1-330 | tee gpurun_out/r04_run15/c3.txt
python tools/bench_configs.py c3 --reps 4 --sweeps 300 --spl 2 2>/dev/null | grep '^{' | cut -c1-330 | tee -a gpurun_out/r04_run15/c3.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_seam.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -3
db() { find "$1" -name '*.db' | head -1; }
cd /tmp; rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d /tmp/c_s -o r -- python $R/tools/bench_configs.py c3 --reps 1 --sweeps 300 > /dev/null 2>&1
python $R/tools/prof_summary.py counters $(db /tmp/c_s) $R/gpurun_out/r04_run15/r04_pmc_sq_issue_c3.txt | grep "k_fused2d<FusedGen2D, 3" | grep "SQ_INSTS_VALU\|SQ_ACTIVE_INST_VALU\|SQ_WAVE_CYCLES" | cut -c1-30,60-130
