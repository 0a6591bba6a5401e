#!/bin/bash
# GPU-side timeline of ONE host-pointer call (kernel trace + memory-copy trace): what runs when, per millisecond.
#   gpurun --timeout 900 -- 'bash tools/host_pipeline_timeline.sh c4 8 500 2'      (config, members, sweeps, host_chunk)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
CFG=${1:-c5}; MEM=${2:-15}; SW=${3:-200}; CH=${4:-0}; BUCKET=${5:-10}
R=$PWD; out=$R/gpurun_out/r06_hp; mkdir -p $out
cd /tmp; rm -rf /tmp/hp_kt
rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/hp_kt -o r -- python $R/tools/bench_host_pipeline.py $CFG --members $MEM --sweeps $SW --chunks $CH --reps 1 > $out/run_$CFG.txt 2>&1
db=$(find /tmp/hp_kt -name '*.db' | head -1)
python - <<PY
import sqlite3, collections, json
c = sqlite3.connect('$db')
wall = json.loads([l for l in open('$out/run_$CFG.txt') if l.startswith('{"entry": "host')][-1])['wall_ms']
rows = c.execute("select name, start, end from kernels order by start").fetchall()
try:
    cps = c.execute("select name, start, end, size from memory_copies order by start").fetchall()
except Exception as e:
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')") if 'cop' in r[0].lower()]
    print('copy tables:', tabs); cps = []
tend = max(rows[-1][2], cps[-1][2] if cps else 0)
span = (wall + 1.0) * 1e6
sel = [r for r in rows if r[1] > tend - span]
csel = [r for r in cps if r[1] > tend - span]
t0 = min(sel[0][1], csel[0][1] if csel else sel[0][1])
B = $BUCKET * 1e5   # bucket in ns (BUCKET in tenths of a millisecond)
b = collections.defaultdict(lambda: [0.0, 0, 0.0, 0.0])
for n, s, e in sel:
    k = int((s - t0) / B); b[k][0] += (e - s) / 1e6; b[k][1] += 1
for n, s, e, sz in csel:
    k = int((s - t0) / B); b[k][2 if 'HOST_TO' in n.upper() or 'H2D' in n.upper() else 3] += sz / 1e6
print('wall of the call %.2f ms; window %.2f ms' % (wall, (tend - t0) / 1e6))
for k in sorted(b): print('t=%6.1f ms  kernels %4d  summed kernel time %6.2f ms   up %7.1f MB  down %7.1f MB' % (k * B / 1e6, b[k][1], b[k][0], b[k][2], b[k][3]))
names = collections.Counter(); dur = collections.Counter()
for n, s, e in sel: names[n[:60]] += 1; dur[n[:60]] += (e - s) / 1e6
for n, v in dur.most_common(10): print('%8.3f ms %5d  %s' % (v, names[n], n))
iv = sorted((s, e) for n, s, e in sel)
busy = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
print('kernels: first %.2f ms, last end %.2f ms, GPU busy (union) %.2f ms, summed %.2f ms' % ((iv[0][0] - t0) / 1e6, (max(e for s, e in iv) - t0) / 1e6, busy / 1e6, sum(e - s for s, e in iv) / 1e6))
for nm in set(n for n, s, e, sz in csel):
    x = [(s, e, sz) for n, s, e, sz in csel if n == nm]
    print('copies %-28s %4d  %.1f MB  first start %.2f ms last end %.2f ms  busy %.2f ms' % (nm, len(x), sum(z for s, e, z in x) / 1e6, (x[0][0] - t0) / 1e6, (max(e for s, e, z in x) - t0) / 1e6, sum(e - s for s, e, z in x) / 1e6))
PY
tail -1 $out/run_$CFG.txt | cut -c1-200
