#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$PWD; out=$R/gpurun_out/r06_hp; mkdir -p $out
cd /tmp; rm -rf /tmp/hp_kt
rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/hp_kt -o r -- python $R/tools/bench_host_pipeline.py c5 --members 15 --sweeps 200 --chunks 0 --reps 1 > $out/run.txt 2>&1
db=$(find /tmp/hp_kt -name '*.db' | head -1)
python - <<PY
import sqlite3
c = sqlite3.connect('$db')
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if 'kern' in t or 'copy' in t or 'memory' in t][:20])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
# the LAST host-pointer call: find the span after the last long gap (> 50 ms) -- simply take the last 400 ms
tend = rows[-1][2]
sel = [r for r in rows if r[1] > tend - 260e6]
t0 = sel[0][1]
import collections
# per 10 ms bucket: busy time of k_pipe3d (sum of durations may exceed wall: two streams), count
b = collections.defaultdict(lambda: [0.0, 0])
for n, s, e in sel:
    k = int((s - t0) / 10e6)
    b[k][0] += (e - s) / 1e6; b[k][1] += 1
for k in sorted(b): print('t=%3d ms  kernels %4d  summed kernel time %.1f ms' % (k * 10, b[k][1], b[k][0]))
names = collections.Counter()
dur = collections.Counter()
for n, s, e in sel:
    names[n[:50]] += 1; dur[n[:50]] += (e - s) / 1e6
for n, v in dur.most_common(8): print('%8.2f ms %5d  %s' % (v, names[n], n))
# union of busy intervals
iv = sorted((s, e) for n, s, e in sel)
busy = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
print('span %.1f ms, GPU busy (union) %.1f ms' % ((iv[-1][1] - t0) / 1e6, busy / 1e6))
p3 = [(s, e) for n, s, e in sel if 'k_pipe3d' in n]
print('k_pipe3d launches %d, mean %.1f us, min %.1f max %.1f' % (len(p3), sum(e - s for s, e in p3) / len(p3) / 1e3, min(e - s for s, e in p3) / 1e3, max(e - s for s, e in p3) / 1e3))
PY
tail -2 $out/run.txt | cut -c1-200
