#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
out=$PWD/gpurun_out/r06_e2e
mkdir -p $out
( timeout 1200 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_f32.py tests/test_gpu_multidev.py tests/test_gpu_large.py tests/test_gpu_plan.py -q -x 2>&1 | tail -4 ) | tee $out/tests2.txt
python bench.py --no-hbm --no-cpu --no-configs 2>$out/bench_err2.txt | grep '^{' > $out/bench_e2e2.json
tail -3 $out/bench_err2.txt
python - <<PY
import json; d=json.load(open('$out/bench_e2e2.json'))
print('value %.4g' % d['value'])
for k, v in d['end_to_end'].items(): print(' e2e', k, v if isinstance(v, str) else {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a != 'workload'})
PY
python tools/bench_frontend.py 2>/dev/null | head -4 | tee $out/frontend2.txt
