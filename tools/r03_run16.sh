#!/bin/bash
# round 3: poll interval + k_skip_norm_tile: suite, bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 2700 python -m pytest tests -m gpu -q -x ) > gpurun_out/r03_gputests_16.txt 2>&1
tail -5 gpurun_out/r03_gputests_16.txt | head -2
for i in 1 2; do python bench.py --no-cpu --no-parity --no-hbm --no-configs 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench default value %.4g active %.4g ms/step %.3f launch %.2f us' % (d['value'], d['value_active'], d['ms_per_step'], d['roofline']['avg_launch_ms']*1e3))"; done
python tools/bench_configs.py c1 c2 c3 c3m c4 c5 --reps 3 2>/dev/null | grep '^{' | python -c "
import json,sys
print(' '.join('%s %.4g' % (json.loads(l)['config'], json.loads(l)['point_sweeps_per_s']) for l in sys.stdin))"
