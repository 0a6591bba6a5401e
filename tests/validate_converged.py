#!/usr/bin/env python3
"""Converged parity for the other BASELINE configurations at full grid size: the GPU result (whatever
path the engine picks) against the reference's lexicographic ordering (oracle, one CPU core).

  python tests/validate_converged.py [c3 c3m c4 c5] [--tol 1e-12]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))   # (this file lives there: it runs the oracle, which only tests/ may)


def main():
    import util
    from xinvert_amd import synthetic
    ap = argparse.ArgumentParser()
    ap.add_argument('configs', nargs='*', default=['c3', 'c4', 'c5'])
    ap.add_argument('--tol', type=float, default=1e-12)
    ap.add_argument('--mx', type=int, default=100000)
    a = ap.parse_args()
    for name in a.configs:
        if name == 'c3':
            p = synthetic.stommel_cartesian(2000, 2000)
        elif name == 'c3m':
            p = synthetic.munk_cartesian(2000, 2000)
        elif name == 'c4':
            p = synthetic.gill_matsuno(720, 1440, 2)
        elif name == 'c5':
            p = synthetic.omega_latlon(50, 360, 720, 1)
        else:
            raise SystemExit('unknown config ' + name)
        q = synthetic.member(p, 0)
        util.run_hip_dev([q], 4, 0.0)
        t = time.time(); S, fl, st = util.run_hip_dev([q], a.mx, a.tol); tg = time.time() - t
        print('%s %s %r  GPU: loops %d, last change %.3e, %.2f s (path %d)'
              % (name, q['kind'], q['S0'].shape, fl[0][2], fl[0][1], tg, st['path']))
        sys.stdout.flush()
        t = time.time(); Sl, fll = util.run_oracle(q, a.mx, a.tol, 0); tc = time.time() - t
        ok = q['coefs'][-1] != util.U
        print('   reference ordering (oracle, 1 core): loops %d, last change %.3e, %.1f s' % (fll[2], fll[1], tc))
        print('   rel-L2 = %.3e, max-norm / max|S| = %.3e, converged solve %.0fx faster'
              % (util.rel_l2(S[0][ok], Sl[ok]), np.abs(S[0][ok] - Sl[ok]).max() / np.abs(Sl[ok]).max(), tc / tg))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
