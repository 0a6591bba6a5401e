#!/bin/bash
# round 3, measurement of the final state: suite, bench lines, profiles, host paths
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -m gpu -q ) > gpurun_out/r03_gputests_final.txt 2>&1
tail -4 gpurun_out/r03_gputests_final.txt
python bench.py 2>/dev/null | grep '^{' > gpurun_out/r03_bench_default.json; python -c "
import json; d=json.load(open('gpurun_out/r03_bench_default.json')); print('bench default value %.4g active %.4g' % (d['value'], d['value_active']))"
python bench.py --config c4 --steps 5 --warmup 2 2>/dev/null | grep '^{' > gpurun_out/r03_bench_c4.json; python -c "
import json; d=json.load(open('gpurun_out/r03_bench_c4.json')); print('bench c4 value %.4g' % d['value'])"
python bench.py --config c5 --steps 3 --warmup 1 2>/dev/null | grep '^{' > gpurun_out/r03_bench_c5.json; python -c "
import json; d=json.load(open('gpurun_out/r03_bench_c5.json')); print('bench c5 value %.4g' % d['value'])"
python bench.py --inproc --no-cpu --no-parity --no-hbm --no-configs --steps 5 2>/dev/null | grep '^{' > gpurun_out/r03_bench_inproc.json
python tools/bench_configs.py c1 c2 c3 c3m c4 c5 c5g ofes --reps 2 2>/dev/null | grep '^{' > gpurun_out/r03_configs.txt
python tools/bench_configs.py c2 --members 8 --reps 2 2>/dev/null | grep '^{' >> gpurun_out/r03_configs.txt
python tools/bench_configs.py c4 --members 32 --reps 2 2>/dev/null | grep '^{' >> gpurun_out/r03_configs.txt
python tools/bench_configs.py c5 --members 15 --reps 2 2>/dev/null | grep '^{' >> gpurun_out/r03_configs.txt
cut -c1-200 gpurun_out/r03_configs.txt
python tools/bench_small_batch.py 2>/dev/null | grep '^{' > gpurun_out/r03_small_batches.txt; cut -c1-200 gpurun_out/r03_small_batches.txt
python tools/bench_frontend.py 2>/dev/null | grep invert_Poisson > gpurun_out/r03_frontend_end_to_end.txt; cat gpurun_out/r03_frontend_end_to_end.txt
( python tools/bench_host_pipeline.py c5 --members 15 --sweeps 200; python tools/bench_host_pipeline.py c4 --members 8 --sweeps 500 ) 2>/dev/null | grep '^{' > gpurun_out/r03_host_pipeline.txt; cut -c1-230 gpurun_out/r03_host_pipeline.txt
bash tools/profile_headline.sh r03 > gpurun_out/r03_profile.log 2>&1; tail -5 gpurun_out/r03_profile.log | cut -c1-200
# C3 (Stommel, Munk) kernel trace + SQ issue counters of the reworked biharmonic kernel
bash tools/r03_run14.sh > gpurun_out/r03_profile_c3m.log 2>&1; tail -12 gpurun_out/r03_profile_c3m.log | cut -c1-160
# k_pipe3d on 15 omega volumes: kernel trace, SQ counters, HBM-side traffic
bash tools/r03_run15.sh > gpurun_out/r03_profile_c5.log 2>&1; tail -6 gpurun_out/r03_profile_c5.log | cut -c1-150
