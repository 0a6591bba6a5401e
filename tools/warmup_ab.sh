#!/bin/bash
# the headline region after 5 (the driver's), 250 and 1000 untimed warm-up steps, process after process on one box
cd "$GRAFT_REPO_ROOT" || exit 1
for w in 5 250; do
  python bench.py --steps 20 --warmup $w --no-configs --no-cpu --no-e2e --no-hbm --no-parity 2>/tmp/err.txt | grep "^{" > /tmp/line.json || { tail -3 /tmp/err.txt; continue; }
  python - <<PY
import json
d = json.load(open('/tmp/line.json')); r = d['roofline']
print('warmup $w value %.4g avg_launch_us %.2f steps %s' % (d['value'], r['avg_launch_ms'] * 1e3, d['step_ms']))
PY
done
