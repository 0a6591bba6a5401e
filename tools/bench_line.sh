#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06_line
python bench.py 2>gpurun_out/r06_line/err.txt | grep '^{' > gpurun_out/r06_line/r06_bench_default.json
python - <<'PY'
import json; d=json.load(open('gpurun_out/r06_line/r06_bench_default.json')); r=d['roofline']
print('value %.4g ms_per_step %.4f avg_launch_ms %.5f frac %.3f traffic %s hbm_frac %.3f' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['traffic'], r['hbm_frac']))
print('sustained', {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d['sustained'].items() if k not in ('note', 'per_second')})
for k, v in d['end_to_end'].items(): print(' e2e', k, v if isinstance(v, str) else {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a != 'workload'})
for c in d['configs']: print(' %-11s %.4g spread %.3f launch %.2f us traffic %s parity %s' % (c['name'], c['value'], c['run_spread'], c['avg_launch_us'], c.get('traffic_bytes_per_point_sweep'), c['parity_bitwise']))
PY
