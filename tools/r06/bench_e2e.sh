#!/bin/bash
# round 6: bench.py with the sustained and end-to-end legs + the host pipeline tool (state before any change to it)
cd "$GRAFT_REPO_ROOT" || exit 1
out=$PWD/gpurun_out/r06_e2e
mkdir -p $out
( time python bench.py 2>$out/bench_err.txt | grep '^{' > $out/bench_default.json ) 2>&1 | grep real
tail -3 $out/bench_err.txt
python - <<PY
import json; d=json.load(open('$out/bench_default.json')); r=d['roofline']
print('value %.4g ms_per_step %.4f avg_launch_ms %.5f traffic %s (%s)' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['traffic'], r.get('traffic_why_null')))
print('sustained', {k: (v if not isinstance(v, list) else ['%.3g' % x for x in v]) for k, v in d['sustained'].items() if k != 'note'})
for k, v in d['end_to_end'].items(): print(' e2e', k, v if isinstance(v, str) else {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a != 'workload'})
for c in d['configs']: print(' %-11s %.4g runs %s spread %.3f launch %.2f us parity %s' % (c['name'], c['value'], ['%.4g' % v for v in c['values']], c['run_spread'], c['avg_launch_us'], c['parity_bitwise']))
PY
python tools/bench_host_pipeline.py c5 --members 15 --sweeps 200 2>/dev/null | grep '^{' | cut -c1-240 | tee $out/host_pipeline_c5.txt
python tools/bench_frontend.py 2>/dev/null | tee $out/frontend.txt
