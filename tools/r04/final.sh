#!/bin/bash
# round 4, measurement of the final state: suite, bench lines, profiles (kernel traces, SQ issue, HBM-side traffic per config)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$PWD
out=$R/gpurun_out/r04_final
mkdir -p $out
( time timeout 2400 python -m pytest tests -m gpu -q ) > $out/r04_gputests_final.txt 2>&1
tail -4 $out/r04_gputests_final.txt
python bench.py 2>$out/bench_err.txt | grep '^{' > $out/r04_bench_default.json; python -c "
import json; d=json.load(open('$out/r04_bench_default.json')); print('bench default value %.4g active %.4g contracted %.4g' % (d['value'], d['value_active'], d['contracted']['value']))
for c in d['configs']: print(' %-11s %.4g members %d alg_frac %.2f parity %s' % (c['name'], c['value'], c['members'], c['alg_frac'], c['parity_bitwise']))"
python bench.py --config c4 --steps 5 --warmup 2 2>/dev/null | grep '^{' > $out/r04_bench_c4.json; python -c "
import json; d=json.load(open('$out/r04_bench_c4.json')); print('bench c4 value %.4g' % d['value'])"
python bench.py --config c5 --steps 3 --warmup 1 2>/dev/null | grep '^{' > $out/r04_bench_c5.json; python -c "
import json; d=json.load(open('$out/r04_bench_c5.json')); print('bench c5 value %.4g' % d['value'])"
python tools/bench_configs.py c1 c2 c3 c3m c4 c5 c5g ofes --reps 2 2>/dev/null | grep '^{' > $out/r04_configs.txt
python tools/bench_configs.py c2 --members 8 --reps 2 2>/dev/null | grep '^{' >> $out/r04_configs.txt
python tools/bench_configs.py c4 --members 64 --reps 2 2>/dev/null | grep '^{' >> $out/r04_configs.txt
python tools/bench_configs.py c5 --members 15 --reps 2 2>/dev/null | grep '^{' >> $out/r04_configs.txt
cut -c1-170 $out/r04_configs.txt
python tools/bench_small_batch.py 2>/dev/null | grep '^{' > $out/r04_small_batches.txt; cut -c1-170 $out/r04_small_batches.txt
bash tools/profile_headline.sh r04 > $out/r04_profile.log 2>&1; tail -5 $out/r04_profile.log | cut -c1-200
cp gpurun_out/prof_r04/*.txt $out/ 2>/dev/null; cp gpurun_out/prof_r04/traffic.json $out/traffic.json
# per-configuration traffic (bench.py's config lines: same workloads and members)
db() { find "$1" -name '*.db' | head -1; }
prof() {  # name kernel-substring bench-prefix members point-sweeps-per-launch command...
  name=$1; ksub=$2; prefix=$3; mem=$4; psl=$5; shift 5
  ( cd /tmp; rm -rf /tmp/c_kt /tmp/c_f /tmp/c_w /tmp/c_s
    rocprofv3 --kernel-trace --stats -d /tmp/c_kt -o r -- "$@" > /dev/null 2>&1
    python $R/tools/prof_summary.py kernels $(db /tmp/c_kt) $out/r04_kernel_trace_$name.txt | head -2 | cut -c1-150
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/c_f -o r -- "$@" > /dev/null 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/c_w -o r -- "$@" > /dev/null 2>&1
    python $R/tools/prof_summary.py counters $(db /tmp/c_f) $out/r04_pmc_fetch_$name.txt > /dev/null
    python $R/tools/prof_summary.py counters $(db /tmp/c_w) $out/r04_pmc_write_$name.txt > /dev/null
    python $R/tools/prof_summary.py config $(db /tmp/c_f) $(db /tmp/c_w) "$ksub" "$name" "$prefix" $mem $psl $out/traffic.json "tools/r04/final.sh: $*"
    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d /tmp/c_s -o r -- "$@" > /dev/null 2>&1
    python $R/tools/prof_summary.py counters $(db /tmp/c_s) $out/r04_pmc_sq_issue_$name.txt | grep "$ksub" | grep "SQ_INSTS_VALU\|SQ_ACTIVE_INST_VALU\|SQ_WAVE_CYCLES" | cut -c1-30,60-130 )
}
prof C3-Stommel "k_fused2d<FusedGen2D, 3" "k_fused2d<FusedGen2D, K=3" 1 12000000 python $R/tools/bench_configs.py c3 --reps 1 --sweeps 300
prof C3-Munk "k_fusedbih" "k_fusedbih" 1 4000000 python $R/tools/bench_configs.py c3m --reps 1 --sweeps 100
# (C4 runs in two lanes: its counters are taken with one, so that a kernel launch is the whole pass)
prof C4 "k_pipe2d<FusedGen2D" "k_pipe2d<Gen2D" 8 33177600 env XINV_LANES=1 python $R/tools/bench_configs.py c4 --members 8 --reps 1 --sweeps 200
prof C1 "k_pipe2d<FusedStd2D" "k_pipe2d<Std2D" 1 259200 python $R/tools/bench_configs.py c1 --reps 1 --sweeps 500
prof C5 "k_pipe3d" "k_pipe3d" 15 388800000 python $R/tools/bench_configs.py c5 --members 15 --reps 1
cat $out/traffic.json | head -60
( python tools/bench_host_pipeline.py c5 --members 15 --sweeps 200; python tools/bench_host_pipeline.py c4 --members 8 --sweeps 500 ) 2>/dev/null | grep '^{' > $out/r04_host_pipeline.txt; cut -c1-230 $out/r04_host_pipeline.txt
