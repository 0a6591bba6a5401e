#!/bin/bash
# kernel trace of a command: the sweep kernel's launches grouped into solves (a gap > 0.5 ms starts a new group): count, mean
# duration, span, summed / span -- resident solves against the host-pointer calls of tools/bench_host_pipeline.py in one trace.
#   gpurun -- 'bash tools/kernel_groups.sh k_pipe2d python tools/bench_host_pipeline.py c4 --members 8 --sweeps 500 --chunks 8,2 --reps 2'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$PWD; K=$1; shift
cd /tmp; rm -rf /tmp/kg
args=(); for a in "$@"; do case "$a" in tools/*|bench.py) a="$R/$a";; esac; args+=("$a"); done
rocprofv3 --kernel-trace -d /tmp/kg -o r -- "${args[@]}" 2>&1 | grep '^{' | cut -c1-200
db=$(find /tmp/kg -name '*.db' | head -1)
python - <<PY
import sqlite3
c = sqlite3.connect('$db')
rows = c.execute("select name, start, end from kernels where name like '%$K%' order by start").fetchall()
groups = []; cur = []
for n, s, e in rows:
    if cur and s - max(x[2] for x in cur) > 5e5: groups.append(cur); cur = []
    cur.append((n, s, e))
if cur: groups.append(cur)
for g in groups:
    span = (max(x[2] for x in g) - g[0][1]) / 1e6
    summed = sum(x[2] - x[1] for x in g) / 1e6
    d = sorted((x[2] - x[1]) / 1e3 for x in g)
    print('launches %5d  mean %6.1f us  median %6.1f  p10 %6.1f  p90 %6.1f   span %7.2f ms  summed %7.2f ms  concurrency %.2f' % (len(g), summed * 1e3 / len(g), d[len(d) // 2], d[len(d) // 10], d[len(d) * 9 // 10], span, summed, summed / span))
PY
