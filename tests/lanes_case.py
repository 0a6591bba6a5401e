"""Child process of tests/test_gpu_lanes.py: solves batches whose members stop at different sweeps, with the sweep loop
cut in <lanes> launch chains (xinv_options.lanes; 0 = the engine's rule), and compares every member with the oracle.
  python tests/lanes_case.py <expected lanes>     prints 'lanes ok: ...' or raises"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import util                                                   # noqa: E402
from oracle import COLOUR_2                                   # noqa: E402


LANES = []


def case(kind, shape, nb, mx, tol, want):
    if kind == 'std3d':
        ps = [util.rand3d(*shape, 'fixed', 'periodic', msk=(m % 2 == 0), seed=10 + m) for m in range(nb)]
    else:
        ps = [util.rand2d(kind, *shape, 'fixed', 'periodic', msk=(m % 2 == 0), seed=10 + m) for m in range(nb)]
    for m, q in enumerate(ps):                                # forcings of very different size: the members stop apart
        q['coefs'][-1] = np.where(q['coefs'][-1] == q['undef'], q['undef'], q['coefs'][-1] * 10.0 ** (-(m % 4)))
    S, fl, st = util.run_hip_dev(ps, mx, tol, lanes=want)
    if want == 0:                                             # the engine's own rule (lane_rule, xinv_hip.hip)
        rate = 6.0e5 if st['pipelined'] else (2.5e5 if kind == 'std3d' else 3.0e5)
        est_us = nb * float(np.prod(shape)) * st['sweeps_per_launch'] / rate
        want = 2 if (nb >= 2 and est_us >= 20.0 and st['path'] == 2) else 1
    want = min(want, nb)
    assert st['lanes'] == want, (kind, shape, nb, st['lanes'], want)
    LANES.append(st['lanes'])
    loops = []
    for m, q in enumerate(ps):
        So, flo = util.run_oracle(q, mx, tol, COLOUR_2)
        assert np.array_equal(S[m], So), (kind, shape, nb, m, 'field differs')
        assert fl[m][2] == flo[2] and fl[m][0] == flo[0] and abs(fl[m][1] - flo[1]) <= 1e-12 * max(1.0, abs(flo[1])), (m, fl[m], flo)
        loops.append(int(flo[2]))
    return loops


def main():
    want = int(sys.argv[1])
    out = []
    # (many members of few workgroups each: in-kernel reducer; the lagged norm needs 32 workgroups per member)
    out.append(case('std2d', (150, 400), 120, 60, 1e-4, want))
    out.append(case('gen2d', (140, 380), 90, 40, 1e-4, want))
    out.append(case('std2d', (700, 1500), 3, 40, 1e-4, want))   # few members of many workgroups: lanes WITH the lagged norm
    out.append(case('std3d', (16, 150, 500), 8, 24, 1e-3, want))
    out.append(case('std3d', (12, 120, 380), 3, 12, 1e-3, want))
    out.append(case('std2d', (384, 1100), 1, 24, 0.0, 1 if want else 0))      # one member: one lane whatever was asked
    assert LANES[0] == (want if want else 2), LANES         # (the first batch is large enough for the rule to cut it)
    assert any(len(set(l)) > 1 for l in out), out             # (some batch really stopped member by member)
    print('lanes ok:', LANES, [(len(l), min(l), max(l)) for l in out])


if __name__ == '__main__':
    main()
