#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -m gpu -q ) > gpurun_out/r03_gputests_4.txt 2>&1
tail -12 gpurun_out/r03_gputests_4.txt
( time python bench.py ) > gpurun_out/r03_bench_2.txt 2>&1
tail -c 400 gpurun_out/r03_bench_2.txt
for mem in 8 64; do python bench.py --config c4 --members $mem --steps 5 --warmup 2 --sweeps 200 2>&1 | grep '^{' > gpurun_out/r03_bench_c4_$mem.json; python -c "
import json,sys; d=json.load(open('gpurun_out/r03_bench_c4_$mem.json')); print('c4 members $mem value %.4g launch %.1f us %s' % (d['value'], d['roofline']['avg_launch_ms']*1e3, d['roofline']['kernel'][:30]))"; done
