#!/usr/bin/env python3
"""Seeded fuzz of the sweep loop in lanes on a GPU box: batches of 2..24 members of every form, sizes
that put the batch in lanes with and without the lagged norm, members that stop on the tolerance at different sweeps --
every member against the oracle, bit for bit.   python tests/fuzz_lanes.py [first_seed] [count] [--single] [--lanes=n]   (xinv_options.lanes = n forces n chains)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import util                                                   # noqa: E402
from oracle import COLOUR_AUTO                                # noqa: E402


LANES = ([int(a.split('=')[1]) for a in sys.argv if a.startswith('--lanes=')] or [0])[0]
SINGLE = '--single' in sys.argv      # one member, 2-D forms, masked-tile lists forced, grids large enough for the lagged norm:
                                     # the single-chain paths (skipped tiles' work on the side stream, three buffers)


def one(seed):
    rng = np.random.default_rng(seed)
    kind = ['std2d', 'gen2d', 'std2dt', 'bih2d', 'std3d', 'gen3d'][int(rng.integers(4 if SINGLE else 6))]
    BCy = ['fixed', 'extend'][int(rng.integers(2))]
    BCx = ['fixed', 'periodic'][int(rng.integers(2))]
    nb = 1 if SINGLE else int(rng.integers(1, 25))            # (one member: the single-chain paths, side stream included)
    uni = int(rng.integers(2))
    if kind in ('std3d', 'gen3d'):
        zc, yc, xc = int(rng.integers(6, 20)), int(rng.integers(12, 60)), int(rng.integers(64, 400))
        mk = (lambda s: util.rand3d(zc, yc, xc, BCy, BCx, 1, seed=s)) if kind == 'std3d' else \
             (lambda s: util.rand3dg(zc, yc, xc, BCy, BCx, 1, seed=s))
        ncu = 3 if kind == 'std3d' else 7
        sh = (zc, yc, xc)
    elif kind == 'bih2d':
        yc, xc = int(rng.integers(30, 200)), 3 * int(rng.integers(30, 300))
        mk = lambda s: util.randbih(yc, xc, BCy, BCx, int(seed & 1), 1, seed=s)
        ncu = 9
        sh = (yc, xc)
    else:
        yc, xc = int(rng.integers(40, 500)), int(rng.integers(64, 1500))
        if SINGLE:
            yc, xc = int(rng.integers(300, 900)), int(rng.integers(900, 2400))
        bnz = int(rng.integers(3) == 0)
        mk = (lambda s: util.rand2dt(yc, xc, BCy, BCx, bnz, 1, seed=s)) if kind == 'std2dt' else \
             (lambda s: util.rand2d(kind, yc, xc, BCy, BCx, bnz, 1, seed=s))
        ncu = 0 if bnz else {'std2d': 3, 'gen2d': 6, 'std2dt': 5}[kind]   # coefficient arrays (the forcing is the last one)
        sh = (yc, xc)
    ps = [mk(seed * 100 + m) for m in range(nb)]
    munk = 0
    if kind == 'bih2d' and not uni:                           # (round 6) the one-pass kernel's vector-stream variants: 0 = all nine
        munk = int(rng.integers(3))                           # arrays vary (VM = 2); 1 = A C D F vary, G H I per row, B = E = 0
        if munk:                                              # (VM = 1); 2 = ... and C holds A's numbers, F holds D's (VM = 3)
            for q in ps:
                q['coefs'] = [np.zeros_like(c) if k in (1, 4) else
                              (np.ascontiguousarray(np.broadcast_to(ps[0]['coefs'][k][:, :1], c.shape)) if k in (6, 7, 8) else c)
                              for k, c in enumerate(q['coefs'])]
                if munk == 2:
                    q['coefs'][2] = q['coefs'][0].copy(); q['coefs'][5] = q['coefs'][3].copy()
    if uni and ncu:                                           # coefficients constant along x, one stack for the batch
        c0 = [np.ascontiguousarray(np.broadcast_to(c[..., :1], c.shape)) for c in ps[0]['coefs'][:ncu]]
        for q in ps:
            q['coefs'][:ncu] = c0
    if SINGLE:                                                # blank blocks of the forcing: whole tiles to skip
        F = ps[0]['coefs'][-1]
        for _ in range(4):
            j0, i0 = int(rng.integers(0, sh[0] // 2)), int(rng.integers(0, sh[1] // 2))
            F[j0:j0 + sh[0] // 3, i0:i0 + sh[1] // 3] = ps[0]['undef']
    for m, q in enumerate(ps):                                # forcings of different size: the members stop apart
        F = q['coefs'][-1]
        q['coefs'][-1] = np.where(F == q['undef'], q['undef'], F * 10.0 ** (-(m % 5)))
    shared = tuple(range(ncu)) if (uni and ncu) else ()
    mx, tol = int(rng.integers(8, 40)), float(10.0 ** rng.uniform(-4, -1.5))
    # opt-in contracted arithmetic where a variant exists (per-row coefficients, 5-point / 7-point, no odd-width seam)
    fma = int(bool(shared) and kind in ('std2d', 'gen2d', 'std3d') and not (BCx == 'periodic' and sh[-1] % 2) and rng.integers(2))
    order = (2 | 0x100) if fma else COLOUR_AUTO
    opt = dict(fma=1) if fma else {}
    if LANES:
        opt['lanes'] = LANES
    hc = int(rng.integers(4))                                 # host-pointer entry: upload / solve / download over member chunks
    if hc:
        opt['host_chunk'] = [0, 1, 3, 7][hc]
    # (round 6) chunk solves in flight / the rolling batch of the standard 3-D form with shared coefficients (-1); the
    # compute-unit count the planner fills (the cut of a launch's last round of k_pipe3d workgroups)
    hi = int(rng.integers(4))
    if hi:
        opt['host_inflight'] = [0, -1, 3, 1][hi]
    if kind == 'std3d' and int(rng.integers(2)):
        opt['cu_count'] = [3, 8, 17, 40][int(rng.integers(4))]
    if kind in ('std2d', 'gen2d', 'std2dt', 'bih2d') and (SINGLE or int(rng.integers(2))):
        opt['force_tile_skip'] = 1                            # masked-tile lists whatever the size
    S, fl, st = util.run_hip_batched(ps, mx, tol, shared=shared, **opt)
    loops = []
    for m, q in enumerate(ps):
        So, flo = util.run_oracle(q, mx, tol, order)
        what = 'seed %d %s %r %s %s nb=%d uni=%d fma=%d member %d lanes=%d path=%d' % (seed, kind, sh, BCy, BCx, nb, uni, fma, m, st['lanes'], st['path'])
        if np.isnan(So).any():
            assert np.array_equal(S[m], So, equal_nan=True), what
        else:
            assert np.array_equal(S[m], So), what + ': %d points differ' % int((S[m] != So).sum())
            assert fl[m][2] == flo[2] and fl[m][0] == flo[0], (what, fl[m], flo)
        loops.append(int(flo[2]))
    return st['lanes'], len(set(loops)) > 1, fma, st['masked_tile_pct'] > 0


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    first = int(args[0]) if len(args) > 0 else 0
    count = int(args[1]) if len(args) > 1 else 100
    bad, laned, apart, nfma, nskip = 0, 0, 0, 0, 0
    for seed in range(first, first + count):
        try:
            l, a, f, sk = one(seed)
            laned += l > 1; apart += a; nfma += f; nskip += sk
        except Exception as e:
            bad += 1
            print('FAIL', str(e)[:400])
    print('lanes fuzz: seeds %d..%d%s, lanes=%s: %d in lanes, %d with members stopping apart, %d contracted, %d with skipped tiles, failures: %d'
          % (first, first + count - 1, ' (single member)' if SINGLE else '', LANES or 'auto', laned, apart, nfma, nskip, bad))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
