#!/usr/bin/env python3
"""GPU-side timeline of short solves on a resident plan (the frame of apps.animate_iteration: 2 sweeps on 73 x 144):
  rocprofv3 --kernel-trace --memory-copy-trace -d DIR -o r -- python tools/r05/frame_timeline.py run
  python tools/r05/frame_timeline.py report DIR/**/r_results.db"""
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run():
    from xinvert_amd import synthetic, resident
    p = synthetic.gill_matsuno(73, 144, 1)
    rp = resident.ResidentProblem(p)
    for _ in range(30):
        rp.solve(2, 1e-12)


def report(db):
    c = sqlite3.connect(db)
    ks = c.execute("select name, start, end from kernels order by start").fetchall()
    try:
        cs = c.execute("select name, start, end from memory_copies order by start").fetchall()
    except Exception:
        cs = []
    ev = sorted([(s, e, n.split('(')[0][:60]) for n, s, e in ks] + [(s, e, 'copy:' + str(n)) for n, s, e in cs])
    starts = [i for i, x in enumerate(ev) if 'k_solve_init' in x[2]]
    i0 = starts[-3]
    t0 = ev[i0][0]
    for s, e, n in ev[i0:]:
        print('+%8.1f us  %6.1f us  %s' % ((s - t0) / 1e3, (e - s) / 1e3, n))
    print('frame period: %.1f us' % ((ev[starts[-1]][0] - ev[starts[-3]][0]) / 2e3))


if __name__ == '__main__':
    run() if sys.argv[1] == 'run' else report(sys.argv[2])
