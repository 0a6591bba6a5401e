#!/bin/bash
# Measurement of a round's final state (RN=r06 by default: files are named ${RN}_*): suite, bench lines, and per configuration -- with the lane count the bench uses --
# a kernel trace, HBM-side traffic (separate FETCH_SIZE / WRITE_SIZE passes), SQ issue counters and the HIP-event
# durations of the same workload (chunk events = what bench.py's `avg_launch` is made of; per-launch events).
#   gpurun --timeout 3000 -- 'bash tools/final.sh'        -> gpurun_out/${RN}_final/*   (copy what is to be judged into profiles/)
# Every traffic.json entry written here carries the hash of the sources it was measured on (tools/prof_summary.py: src_sha);
# bench.py reports an entry only while that hash matches the tree it runs from.
cd "$GRAFT_REPO_ROOT" || exit 1
RN=${RN:-r06}
export TMPDIR=/tmp
R=$PWD
out=$R/gpurun_out/${RN}_final
mkdir -p $out
if [ "$1" != "noprof" ]; then
( time timeout 2400 python -m pytest tests -m gpu -q ) > $out/${RN}_gputests_final.txt 2>&1
tail -4 $out/${RN}_gputests_final.txt
fi
python bench.py 2>$out/bench_err.txt | grep '^{' > $out/${RN}_bench_default.json; python -c "
import json; d=json.load(open('$out/${RN}_bench_default.json')); r=d['roofline']
print('bench default value %.4g active %.4g contracted %.4g ms_per_step %.4f launches x avg %.4f outside %.4f' % (d['value'], d['value_active'], d['contracted']['value'], d['ms_per_step'], r['launches']/d['steps']*r['avg_launch_ms'], r['ms_outside_launches']))
print(' roofline first keys:', list(r.keys())[:24])
print(' hbm leg frac %.3f traffic %s' % (d['roofline_hbm']['frac'], d['roofline_hbm'].get('traffic')))
for c in d['configs']: print(' %-11s %.4g members %d launch %.2f us alg_frac %.2f traffic %s parity %s' % (c['name'], c['value'], c['members'], c['avg_launch_us'], c['alg_frac'], c.get('traffic_bytes_per_point_sweep'), c['parity_bitwise']))"
python bench.py --config c4 --steps 5 --warmup 2 2>/dev/null | grep '^{' > $out/${RN}_bench_c4.json; python -c "
import json; d=json.load(open('$out/${RN}_bench_c4.json')); print('bench c4 value %.4g' % d['value'])"
python bench.py --config c5 --steps 3 --warmup 1 2>/dev/null | grep '^{' > $out/${RN}_bench_c5.json; python -c "
import json; d=json.load(open('$out/${RN}_bench_c5.json')); print('bench c5 value %.4g' % d['value'])"
python tools/bench_configs.py c1 gm73 c2 c3 c3m c3mxy c4 c5 c5g ofes --reps 2 2>/dev/null | grep '^{' > $out/${RN}_configs.txt
python tools/bench_configs.py c2 --members 8 --reps 2 2>/dev/null | grep '^{' >> $out/${RN}_configs.txt
python tools/bench_configs.py c4 --members 64 --reps 2 2>/dev/null | grep '^{' >> $out/${RN}_configs.txt
python tools/bench_configs.py c5 --members 15 --reps 2 2>/dev/null | grep '^{' >> $out/${RN}_configs.txt
cut -c1-200 $out/${RN}_configs.txt
python tools/bench_animate.py 2>/dev/null | grep "^{" > $out/${RN}_animate.txt; cat $out/${RN}_animate.txt
python tools/bench_small_batch.py 2>/dev/null | grep '^{' > $out/${RN}_small_batches.txt; cut -c1-170 $out/${RN}_small_batches.txt
for i in 1 2 3; do python tools/solve_overhead.py 2>/dev/null | grep '^{'; done > $out/${RN}_solve_overhead_final.txt; python tools/solve_overhead.py --plan 0 2>/dev/null | grep '^{' >> $out/${RN}_solve_overhead_final.txt; cat $out/${RN}_solve_overhead_final.txt | cut -c1-260
[ "$1" = "noprof" ] && exit 0
cp $R/profiles/traffic.json $out/traffic.json 2>/dev/null
db() { find "$1" -name '*.db' | head -1; }
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"
# name, kernel substring, bench.py kernel prefix, members, lanes, point-sweeps per KERNEL LAUNCH (= members / lanes x points x K), command...
prof() {
  name=$1; ksub=$2; prefix=$3; mem=$4; lanes=$5; psl=$6; shift 6
  ( cd /tmp; rm -rf /tmp/c_kt /tmp/c_f /tmp/c_w /tmp/c_s
    # the unprofiled HIP-event durations of the very same command (what bench.py's avg_launch is made of), then the profiles
    "$@" --events 2>/dev/null | grep '^{' > $out/${RN}_launch_events_$name.txt
    rocprofv3 --kernel-trace --stats -d /tmp/c_kt -o r -- "$@" > /dev/null 2>&1
    python $R/tools/prof_summary.py kernels $(db /tmp/c_kt) $out/${RN}_kernel_trace_$name.txt | head -2 | cut -c1-150
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/c_f -o r -- "$@" > /dev/null 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/c_w -o r -- "$@" > /dev/null 2>&1
    python $R/tools/prof_summary.py counters $(db /tmp/c_f) $out/${RN}_pmc_fetch_$name.txt > /dev/null
    python $R/tools/prof_summary.py counters $(db /tmp/c_w) $out/${RN}_pmc_write_$name.txt > /dev/null
    python $R/tools/prof_summary.py config $(db /tmp/c_f) $(db /tmp/c_w) "$ksub" "$name" "$prefix" $mem $psl $out/traffic.json "tools/final.sh ($lanes lane(s); a kernel launch covers members / lanes): $*"
    rocprofv3 --kernel-trace --pmc $SQ -d /tmp/c_s -o r -- "$@" > /dev/null 2>&1
    python $R/tools/prof_summary.py counters $(db /tmp/c_s) $out/${RN}_pmc_sq_issue_$name.txt | grep "$ksub" | grep "SQ_INSTS_VALU\|SQ_ACTIVE_INST_VALU\|SQ_WAVE_CYCLES" | cut -c1-30,60-130
    python - <<EOF
import json
e = json.loads(open('$out/${RN}_launch_events_$name.txt').read().strip().splitlines()[-1])
kt = open('$out/${RN}_kernel_trace_$name.txt').read().splitlines()
row = [l for l in kt if '$ksub'[:40] in l][:1]
with open('$out/${RN}_launch_events_$name.txt', 'a') as f:
    f.write('# the same command under rocprofv3 --kernel-trace (${RN}_kernel_trace_$name.txt): ' + (row[0][:110] if row else '?') + '\n')
    f.write('# HIP events, unprofiled: %d passes x %.2f us = %.3f ms of a %.3f ms solve (%d lane(s): with two, a pass is two overlapping kernel launches)\n'
            % (e['launches'], e['avg_launch_ms'] * 1e3, e['launches_x_avg_ms'], e['solve_ms'], e['lanes']))
EOF
  )
}
prof C2 "k_pipe2d<FusedStd2D" "k_pipe2d<Std2D" 1 1 25920000 python $R/tools/bench_configs.py c2 --reps 4 --sweeps 500
prof C3-Stommel "k_fused2d<FusedGen2DQ_<true>, 3" "k_fused2d<FusedGen2D, K=3" 1 1 12000000 python $R/tools/bench_configs.py c3 --reps 4 --sweeps 300
prof C3-Munk "k_fusedbih<false, true, 0>" "k_fusedbih (one pass per sweep, A..I" 1 1 4000000 python $R/tools/bench_configs.py c3m --reps 4 --sweeps 100
prof C3-Munk-xy "k_fusedbih<false, true, 3>" "k_fusedbih (one pass per sweep; A (= C), D (= F)" 1 1 4000000 python $R/tools/bench_configs.py c3mxy --reps 4 --sweeps 100
prof C4 "k_pipe2d<FusedGen2D" "k_pipe2d<Gen2D" 8 2 16588800 python $R/tools/bench_configs.py c4 --members 8 --reps 4 --sweeps 200
prof C1 "k_pipe2d<FusedStd2D" "k_pipe2d<Std2D" 1 1 259200 python $R/tools/bench_configs.py c1 --reps 4 --sweeps 500
prof C5 "k_pipe3d" "k_pipe3d" 15 1 388800000 python $R/tools/bench_configs.py c5 --members 15 --reps 2
# the headline workload through bench.py itself (kernel trace + traffic of the HBM leg: 8 members in two lanes)
cmd="python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-parity --no-configs --no-e2e"
( cd /tmp; rm -rf /tmp/p_kt /tmp/p_f /tmp/p_w
  rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o r -- $cmd > /dev/null 2>&1
  python $R/tools/prof_summary.py kernels $(db /tmp/p_kt) $out/${RN}_kernel_trace_bench.txt | head -4 | cut -c1-150
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_f -o r -- $cmd > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_w -o r -- $cmd > /dev/null 2>&1
  python $R/tools/prof_summary.py counters $(db /tmp/p_f) $out/${RN}_pmc_fetch_bench.txt > /dev/null
  python $R/tools/prof_summary.py counters $(db /tmp/p_w) $out/${RN}_pmc_write_bench.txt > /dev/null
  python $R/tools/prof_summary.py traffic $(db /tmp/p_f) $(db /tmp/p_w) "k_pipe2d<FusedStd2D, 3u, false, 1" std2d_pipe_um3 $out/traffic.json 1 1
  python $R/tools/prof_summary.py traffic $(db /tmp/p_f) $(db /tmp/p_w) "k_fused2d<FusedStd2D, 1" std2d_spl1_um0_all $out/traffic.json 8 2 )
python -c "
import json; d=json.load(open('$out/traffic.json')); print({k: (v if not isinstance(v, dict) else '...') for k, v in d.items() if not k.endswith('_detail') and k != 'configs'}); print({k: round(v['bytes_per_point_sweep'], 2) for k, v in d['configs'].items()}); print(d.get('std2d_spl1_um0_all_detail'))"
( python tools/bench_host_pipeline.py c5 --members 15 --sweeps 200; python tools/bench_host_pipeline.py c4 --members 8 --sweeps 500 ) 2>/dev/null | grep '^{' > $out/${RN}_host_pipeline_final.txt; cut -c1-230 $out/${RN}_host_pipeline_final.txt
