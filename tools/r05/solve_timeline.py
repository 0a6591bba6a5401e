#!/usr/bin/env python3
"""What a headline solve spends outside its sweep launches: timeline of one solve from a rocprofv3 kernel + memory-copy trace.
  rocprofv3 --kernel-trace --memory-copy-trace -d DIR -o r -- python tools/r05/solve_timeline.py run
  python tools/r05/solve_timeline.py report DIR/**/r_results.db"""
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run():
    import numpy as np
    from xinvert_amd import synthetic, resident
    p = synthetic.poisson_latlon(1800, 3600, mask=True, members=1)
    rp = resident.ResidentProblem(p)
    for _ in range(6):
        rp.reset(); rp.solve(499, 0.0)


def report(db):
    c = sqlite3.connect(db)
    ks = c.execute("select name, start, end from kernels order by start").fetchall()
    try:
        cs = c.execute("select name, start, end from memory_copies order by start").fetchall()
    except Exception:
        cs = []
    ev = sorted([(s, e, n) for n, s, e in ks] + [(s, e, 'copy:' + str(n)) for n, s, e in cs])
    # the last solve: from the last k_solve_init to the end (round 5: solves run on a resident plan -- no detection pass)
    starts = [i for i, x in enumerate(ev) if 'k_solve_init' in x[2]]
    while starts and ev[starts[-1] - 1][2].startswith('copy') is False and 'fill' in ev[starts[-1] - 1][2].lower():
        starts[-1] -= 1                                   # (the memset of the partials precedes it)
    i0 = starts[-1] if starts else 0
    seg = ev[i0:]
    t0, t1 = seg[0][0], max(x[1] for x in seg)
    busy = {}
    for s, e, n in seg:
        key = n.split('(')[0][:50]
        busy[key] = busy.get(key, 0) + (e - s)
    print('last solve: span %.1f us, %d events' % ((t1 - t0) / 1e3, len(seg)))
    for k, v in sorted(busy.items(), key=lambda kv: -kv[1]):
        print('  %-52s %9.1f us' % (k, v / 1e3))
    print('  %-52s %9.1f us' % ('(idle between events)', ((t1 - t0) - sum(busy.values())) / 1e3))
    # the events before the first sweep launch and after the last
    first = next(i for i, x in enumerate(seg) if 'k_pipe2d' in x[2])
    last = max(i for i, x in enumerate(seg) if 'k_pipe2d' in x[2] or 'k_fused2d' in x[2])
    print('before the first sweep launch: %.1f us; after the last: %.1f us' % ((seg[first][0] - t0) / 1e3, (t1 - seg[last][1]) / 1e3))
    for s, e, n in seg[:first]:
        print('    +%8.1f us  %7.1f us  %s' % ((s - t0) / 1e3, (e - s) / 1e3, n[:70]))
    for s, e, n in seg[last + 1:]:
        print('    +%8.1f us  %7.1f us  %s' % ((s - t0) / 1e3, (e - s) / 1e3, n[:70]))
    gaps = [(seg[i + 1][0] - seg[i][1]) / 1e3 for i in range(first, last)]
    big = sorted(gaps)[-8:]
    print('gaps between sweep-phase events: mean %.2f us, total %.1f us, largest %s' % (sum(gaps) / len(gaps), sum(gaps), ['%.1f' % g for g in big]))


if __name__ == '__main__':
    run() if sys.argv[1] == 'run' else report(sys.argv[2])
