#!/bin/bash
# round 3, call 2: graph variants of the scalar-cache probe; general-form pipelined pass: parity + C4 A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time ./build/scache_probe 1500 ) > gpurun_out/r03_scache_probe.txt 2>&1
grep -v "variant [0-3]" gpurun_out/r03_scache_probe.txt | head -40
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_scalar_cache.py -m gpu -x -q -k "pipelined or c4 or c1 or gen or alternating" ) > gpurun_out/r03_gputests_2.txt 2>&1
tail -8 gpurun_out/r03_gputests_2.txt
for mode in 1 0; do
  for mem in 8 64; do
    echo "== XINV_PIPE=$mode c4 members $mem"
    XINV_PIPE=$mode python bench.py --config c4 --members $mem --steps 5 --warmup 2 --sweeps 200 2>&1 | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); r = d['roofline']
        print('value %.4g  launch %.1f us  kernel %s  K %d rows %d  bound %s frac %.3f valu %.3f streamed %.3f' % (d['value'], r['avg_launch_ms']*1e3, r['kernel'], d['config']['sweeps_per_launch'], d['config']['rows_per_tile'], r['bound'], r['frac'], r['valu_frac'], r['streamed_frac_of_hbm_peak'] or 0))
    elif 'rror' in ln: print(ln.rstrip())
"
  done
done 2>&1 | tee gpurun_out/r03_c4_ab.txt
