#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/q_tl -o r -- python $R/bench.py --steps 1 --warmup 1 --no-cpu --no-parity --no-hbm --no-configs > /dev/null 2>&1
python $R/tools/prof_timeline.py $(find /tmp/q_tl -name '*.db' | head -1) 6 | tail -90
