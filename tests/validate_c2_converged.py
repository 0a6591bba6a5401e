#!/usr/bin/env python3
"""north_star's acceptance criterion at the headline size: invert_Poisson 3600 x 1800 (land mask) run to
convergence on the GPU (red-black, K sweeps per pass, masked tiles skipped) against the reference's lexicographic
ordering (oracle, one CPU core, ~2.5 minutes).  Prints loops, wall times and the rel-L2 difference."""
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
    tol = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-12
    p = synthetic.poisson_latlon(1800, 3600, mask=True)
    q = synthetic.member(p, 0)
    util.run_hip_dev([q], 10, 0.0)
    t = time.time(); S, fl, st = util.run_hip_dev([q], 200000, tol); tg = time.time() - t
    print('GPU   : loops %d, last relative change %.3e, %.2f s (path %d, K=%d, %d%% of the tiles skipped)'
          % (fl[0][2], fl[0][1], tg, st['path'], st['sweeps_per_launch'], st['masked_tile_pct']))
    sys.stdout.flush()
    t = time.time(); Sl, fll = util.run_oracle(q, 200000, tol, 0); tc = time.time() - t
    print('oracle: loops %d, last relative change %.3e, %.1f s (lexicographic = reference ordering, 1 core)'
          % (fll[2], fll[1], tc))
    ok = q['coefs'][3] != util.U
    print('rel-L2(GPU - reference ordering) over sea points = %.3e   (north_star: <= 1e-6)' % util.rel_l2(S[0][ok], Sl[ok]))
    print('max |diff| / max |psi| = %.3e' % (np.abs(S[0][ok] - Sl[ok]).max() / np.abs(Sl[ok]).max()))
    print('speed-up of the converged solve: %.0fx' % (tc / tg))


if __name__ == '__main__':
    main()
