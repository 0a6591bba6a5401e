"""Tolerances within ONE ULP of a sweep's relative change, on every norm path.

  python tests/stop_rule_edge.py c2|c5 [out.txt]        (XINV_LAG=0 in the environment: the in-kernel reducer)

The stop rule is `flags[1] = |norm - normPrev| / normPrev < tolerance` (numbas.py:401-414) with norm = mean|S|.  The
sweeps are bitwise the oracle's on every path, but the norm's SUMMATION ORDER is the path's own (the oracle: row-major;
the streaming kernels: per-tile partials added by a reducer; the watchdog recovery and the colour launches:
k_norm_partial / k_norm_final), so r_k agrees to rounding only, and a tolerance within an ulp of r_k may or may not
stop a given path at sweep k.  What MUST hold, and is asserted here:
  * with a tolerance a relative 1e-9 away from r_k on either side, every path stops exactly where the oracle does;
  * with a tolerance within one ulp of r_k, a path stops at sweep k or at the next sweep that meets the tolerance
    (k + 1 here) -- nowhere else --, the field it returns is BITWISE the oracle's field after that many sweeps, and its
    flags[1] is the oracle's r of that sweep to 1e-12.
Which side each path lands on is printed (and kept in profiles/).  Test infrastructure: the oracle is the checker."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import util                                                # noqa: E402
import oracle as orc                                       # noqa: E402
from xinvert_amd import _lib, synthetic                    # noqa: E402


def problem(name):
    if name == 'c2':
        p = synthetic.poisson_latlon(1800, 3600, mask=True)
    else:
        p = synthetic.omega_latlon(50, 360, 720, steps=1)
    return synthetic.member(p, 0), p['shared']


def main():
    name = sys.argv[1]
    out = open(sys.argv[2], 'a') if len(sys.argv) > 2 else None
    _lib.require_gpu()
    q, shared = problem(name)
    lag = os.environ.get('XINV_LAG', '1') != '0'
    # the oracle's history around sweep k: states and relative changes
    k = 11
    hist = {}
    for kk in (k - 1, k, k + 1, k + 2):
        hist[kk] = util.run_oracle(q, kk, 0.0, orc.COLOUR_2)
    r = {kk: hist[kk][1][1] for kk in hist}
    assert r[k - 1] > r[k] > r[k + 1] > r[k + 2] > 0, 'pick a sweep where the change falls monotonically: %r' % r
    ulp = np.spacing(r[k])
    cases = [('r_k (1 - 1e-9)', r[k] * (1 - 1e-9), {k + 1}), ('r_k - 1 ulp', r[k] - ulp, {k, k + 1}), ('r_k', r[k], {k, k + 1}),
             ('r_k + 1 ulp', r[k] + ulp, {k, k + 1}), ('r_k (1 + 1e-9)', r[k] * (1 + 1e-9), {k})]
    paths = [('lagged reducer' if lag else 'in-kernel reducer', {}, None),
             ('watchdog recovery from launch 1', {}, '1,0'),
             ('colour launches', dict(path=1), None)]
    if name == 'c5':
        paths.insert(1, ('one sweep per pass', dict(sweeps_per_launch=1), None))
    else:
        paths.insert(1, ('k_fused2d (no pipeline)', dict(no_pipe=1), None))
    bad = 0
    for pname, opt, wd in paths:
        for cname, tol, allowed in cases:
            if wd:
                os.environ['XINV_EXP_WATCHDOG'] = wd
            try:
                S, fl, st = util.run_hip_dev([q], 400, float(tol), shared=shared, **opt)
            finally:
                os.environ.pop('XINV_EXP_WATCHDOG', None)
            loop = int(fl[0][2])
            ok = loop in allowed and np.array_equal(S[0], hist[loop][0]) and abs(fl[0][1] - r[loop]) <= 1e-12 \
                and (wd is None or st['recovered_members'] == 1)
            bad += 0 if ok else 1
            line = '%s  %-32s tolerance = %-16s stops at sweep %d (oracle: r_k = %.17g at k = %d; r here %.17g, %+d ulp)%s' % (
                name, pname, cname, loop, r[k], k, fl[0][1], int(round((fl[0][1] - r[loop]) / np.spacing(r[loop]))), '' if ok else '   <-- WRONG')
            print(line, flush=True)
            if out:
                out.write(line + '\n')
    print('%s: %d wrong' % (name, bad))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
