#!/bin/bash
# round 6: the state of the tree at the start of the round on this round's boxes (A/B reference for what follows)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=$PWD/gpurun_out/r06_base
mkdir -p $out
python bench.py 2>$out/bench_err.txt | grep '^{' > $out/bench_default.json
python - <<PY
import json; d=json.load(open('$out/bench_default.json')); r=d['roofline']
print('value %.4g ms_per_step %.4f avg_launch_ms %.5f outside %.4f' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['ms_outside_launches']))
for c in d['configs']: print(' %-11s %.4g members %d launch %.2f us parity %s' % (c['name'], c['value'], c['members'], c['avg_launch_us'], c['parity_bitwise']))
PY
for m in 13 14 15 16; do python tools/bench_configs.py c5 --members $m --reps 2 2>/dev/null | grep '^{' | cut -c1-400; done > $out/c5_members.txt
cat $out/c5_members.txt
