"""Reuse stress for one kernel family, in a FRESH process (so the first launch of its kernel variants happens
here): two different problems of one geometry alternate through the same workspace and device addresses --
A, B, A, B, ... -- and every solve must equal the oracle bit for bit in S, with the oracle's loop index.

  python tests/stress_scalar_cache.py FAMILY [alternations]

A kernel that served coefficient rows, per-row factors, control-block words, tile lists or the skipped tiles'
norm share of the PREVIOUS solve from a stale cache (the scalar data cache is not coherent with vector stores:
DESIGN.md 4.1c) relaxes B with A's data and fails here.  The host-pointer entry is used: its device pool
hands the same addresses to every call, so each array is rewritten in place between solves.
Test infrastructure: the oracle is the checker, the product path is the HIP library.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import util                                                # noqa: E402
import oracle as orc                                       # noqa: E402
from xinvert_amd import _lib                               # noqa: E402

U = util.U


def xuni(p, idx):
    """make coefficient arrays `idx` constant along x (what a lat-lon builder hands over)"""
    for k in idx:
        c = p['coefs'][k]
        c[...] = c[..., :1]
    return p


def pair(make):
    """two problems of one geometry with different coefficients, forcing and first guess"""
    return make(101), make(202)


# family -> (builder(seed) -> problem, engine options, expected stats, oracle ordering)
def families():
    F = {}
    # lat-lon standard form, A and C per row: the wave-pipelined four-sweep pass (per-row factor records + tile list
    # through the scalar unit) -- with the masked-tile lists and with the lagged norm
    F['pipe2d'] = (lambda s: xuni(util.rand2d('std2d', 96, 384, 'fixed', 'periodic', seed=s), (0, 2)),
                   {}, dict(path=2, pipelined=1, sweeps_per_launch=4, xuniform_mask=3), orc.COLOUR_2)
    F['pipe2d_ext'] = (lambda s: xuni(util.rand2d('std2d', 96, 384, 'extend', 'periodic', seed=s), (0, 2)),
                       {}, dict(path=2, pipelined=1), orc.COLOUR_2)
    F['pipe2d_skip'] = (lambda s: masked_blocks(xuni(util.rand2d('std2d', 384, 768, 'fixed', 'periodic', seed=s), (0, 2)), s),
                        dict(force_tile_skip=1), dict(path=2, pipelined=1, masked_min=1), orc.COLOUR_2)
    # the same with the forcing riding the LDS ring (the variant large batches take; forced here: xinv_options.pipe_fr = 1)
    F['pipe2d_fr'] = (lambda s: xuni(util.rand2d('std2d', 96, 384, 'extend', 'periodic', msk=True, seed=s), (0, 2)),
                      dict(pipe_fr=1), dict(path=2, pipelined=1), orc.COLOUR_2)
    F['pipe2d_gen_fr'] = (lambda s: xuni(util.rand2d('gen2d', 96, 384, 'fixed', 'periodic', msk=True, seed=s), (0, 2, 3, 4, 5)),
                          dict(pipe_fr=1), dict(path=2, pipelined=1, xuniform_mask=31), orc.COLOUR_2)
    F['fused2d_std_um3'] = (lambda s: xuni(util.rand2d('std2d', 96, 384, 'fixed', 'periodic', seed=s), (0, 2)),
                            dict(no_pipe=1), dict(path=2, pipelined=0, sweeps_per_launch=4, xuniform_mask=3), orc.COLOUR_2)
    F['fused2d_std_um3_skip'] = (lambda s: masked_blocks(xuni(util.rand2d('std2d', 384, 768, 'extend', 'fixed', seed=s), (0, 2)), s),
                                 dict(no_pipe=1, force_tile_skip=1), dict(path=2, pipelined=0, xuniform_mask=3, masked_min=1), orc.COLOUR_2)
    F['fused2d_std_full'] = (lambda s: util.rand2d('std2d', 96, 384, 'fixed', 'fixed', msk=True, seed=s),
                             {}, dict(path=2, xuniform_mask=0), orc.COLOUR_2)
    F['fused2d_gen_um31'] = (lambda s: xuni(util.rand2d('gen2d', 96, 384, 'fixed', 'periodic', seed=s), (0, 2, 3, 4, 5)),
                             dict(no_pipe=1), dict(path=2, xuniform_mask=31, pipelined=0), orc.COLOUR_2)
    F['pipe2d_gen'] = (lambda s: xuni(util.rand2d('gen2d', 96, 384, 'extend', 'periodic', seed=s), (0, 2, 3, 4, 5)),
                       {}, dict(path=2, xuniform_mask=31, pipelined=1, sweeps_per_launch=4), orc.COLOUR_2)
    F['fused2d_gen_um28'] = (lambda s: xuni(util.rand2d('gen2d', 96, 384, 'extend', 'fixed', seed=s), (3, 4, 5)),
                             {}, dict(path=2, xuniform_mask=28), orc.COLOUR_2)
    F['fused2d_gen_full'] = (lambda s: util.rand2d('gen2d', 96, 384, 'fixed', 'fixed', msk=True, seed=s),
                             {}, dict(path=2, xuniform_mask=0), orc.COLOUR_2)
    F['fused2d_std2dt_um7'] = (lambda s: xuni(util.rand2dt(96, 384, 'fixed', 'periodic', seed=s), (0, 3, 4)),
                               {}, dict(path=2, xuniform_mask=7), orc.COLOUR_2)
    F['fused9_std'] = (lambda s: util.rand2d('std2d', 96, 384, 'fixed', 'periodic', bnz=True, msk=True, seed=s),
                       {}, dict(path=2, colours=4), orc.COLOUR_AUTO)
    F['fused9_gen'] = (lambda s: util.rand2d('gen2d', 96, 384, 'extend', 'fixed', bnz=True, seed=s),
                       {}, dict(path=2, colours=4), orc.COLOUR_AUTO)
    F['fused3d_uni'] = (lambda s: xuni(util.rand3d(12, 40, 256, 'fixed', 'periodic', seed=s), (0, 1, 2)),
                        {}, dict(path=2, xuniform_mask=7), orc.COLOUR_2)
    F['fused3d_full'] = (lambda s: util.rand3d(12, 40, 256, 'extend', 'fixed', msk=True, seed=s),
                         {}, dict(path=2, xuniform_mask=0), orc.COLOUR_2)
    F['fused3d_two_sweeps'] = (lambda s: xuni(util.rand3d(12, 40, 256, 'fixed', 'periodic', seed=s), (0, 1, 2)),
                               dict(sweeps_per_launch=2), dict(path=2, sweeps_per_launch=2), orc.COLOUR_2)
    F['fused3dg'] = (lambda s: xuni(util.rand3dg(12, 40, 256, 'fixed', 'periodic', seed=s), range(7)),
                     {}, dict(path=2, xuniform_mask=127), orc.COLOUR_2)
    F['fusedbih'] = (lambda s: xuni(util.randbih(96, 384, 'fixed', 'fixed', bnz=True, seed=s), range(9)),
                     {}, dict(path=2), orc.COLOUR_AUTO)
    F['fusedbih_ext_per'] = (lambda s: xuni(util.randbih(96, 384, 'extend', 'periodic', seed=s), range(9)),
                             {}, dict(path=2), orc.COLOUR_AUTO)
    F['bih_rowclass_uni'] = (lambda s: xuni(util.randbih(96, 384, 'fixed', 'fixed', bnz=True, seed=s), (0, 2, 3)),
                             dict(path=1), dict(path=1), orc.COLOUR_AUTO)
    # the one-pass kernel's vector-stream variants (round 6): vm = 1 reads G, H, I out of the per-row records (scalar unit)
    # beside the A, C, D, F streams; vm = 2 reads every coefficient as a vector
    def munk_like(p):
        p = xuni(p, (6, 7, 8))
        p['coefs'] = [np.zeros_like(c) if k in (1, 4) else c for k, c in enumerate(p['coefs'])]
        return p
    F['fusedbih_vm1'] = (lambda s: munk_like(util.randbih(96, 384, 'fixed', 'fixed', bnz=True, seed=s)),
                         {}, dict(path=2, point_factor=1), orc.COLOUR_AUTO)
    F['fusedbih_vm2'] = (lambda s: util.randbih(96, 384, 'extend', 'periodic', bnz=True, seed=s),
                         {}, dict(path=2, point_factor=2), orc.COLOUR_AUTO)
    F['bih_colour_uni'] = (lambda s: xuni(util.randbih(60, 250, 'extend', 'periodic', bnz=True, seed=s), (0, 2, 3)),
                           {}, dict(path=1), orc.COLOUR_AUTO)
    # colour-pass kernels (odd-xc periodic seam; 'extend' runs k_extend, whose corner reads are block-uniform)
    F['colour_std2d_ext'] = (lambda s: util.rand2d('std2d', 60, 251, 'extend', 'periodic', msk=True, seed=s),
                             dict(path=1), dict(path=1), orc.COLOUR_AUTO)
    # the odd-xc periodic seam inside the streaming kernels (round 4): full arrays, and per-row A and C
    F['fused2d_seam'] = (lambda s: util.rand2d('std2d', 60, 251, 'extend', 'periodic', msk=True, seed=s),
                         {}, dict(path=2), orc.COLOUR_2)
    F['fused2d_seam_um3'] = (lambda s: xuni(util.rand2d('std2d', 96, 385, 'fixed', 'periodic', seed=s), (0, 2)),
                             {}, dict(path=2, xuniform_mask=3), orc.COLOUR_2)
    F['colour_gen2d_nine'] = (lambda s: util.rand2d('gen2d', 60, 251, 'extend', 'periodic', bnz=True, seed=s),
                              dict(path=1), dict(path=1), orc.COLOUR_AUTO)
    F['fused9_seam'] = (lambda s: util.rand2d('std2d', 60, 251, 'extend', 'periodic', bnz=True, msk=True, seed=s),
                        {}, dict(path=2, colours=6), orc.COLOUR_AUTO)
    F['colour_std3d_ext'] = (lambda s: util.rand3d(9, 30, 121, 'extend', 'periodic', seed=s),
                             dict(path=1), dict(path=1), orc.COLOUR_AUTO)
    # the seam inside k_fused3d (ring layout, full coefficient arrays: the seam lanes' east coefficient from the next lane)
    F['fused3d_seam'] = (lambda s: util.rand3d(9, 30, 121, 'extend', 'periodic', msk=True, seed=s),
                         {}, dict(path=2, xuniform_mask=0), orc.COLOUR_2)
    F['fused3dg_seam'] = (lambda s: xuni(util.rand3dg(12, 40, 257, 'extend', 'periodic', seed=s), range(7)),
                          {}, dict(path=2, xuniform_mask=127), orc.COLOUR_2)
    F['fused3d_seam_uni'] = (lambda s: xuni(util.rand3d(12, 40, 257, 'fixed', 'periodic', seed=s), (0, 1, 2)),
                             dict(sweeps_per_launch=1), dict(path=2, xuniform_mask=7, sweeps_per_launch=1), orc.COLOUR_2)
    # the seam as an even ring with a phantom column (round 5): k_pipe3d's two-sweep pass, k_pipe2d's four-sweep pass
    F['fused3d_seam_ring'] = (lambda s: xuni(util.rand3d(12, 40, 257, 'fixed', 'periodic', seed=s), (0, 1, 2)),
                        {}, dict(path=2, xuniform_mask=7, sweeps_per_launch=2), orc.COLOUR_2)
    F['pipe2d_seam'] = (lambda s: xuni(util.rand2d('std2d', 96, 385, 'extend', 'periodic', msk=True, seed=s), (0, 2)),
                        {}, dict(path=2, xuniform_mask=3, sweeps_per_launch=4, pipelined=1), orc.COLOUR_2)
    return F


def masked_blocks(p, seed):
    """blank whole blocks of the forcing so that wave-tiles are skipped (different blocks for each problem)"""
    rng = np.random.default_rng(seed + 7)
    F = p['coefs'][-1]
    yc, xc = F.shape
    for _ in range(6):
        j0, i0 = rng.integers(0, yc - 100), rng.integers(0, xc - 260)
        F[j0:j0 + 100, i0:i0 + 260] = U
    return p


def main():
    fam = sys.argv[1]
    nalt = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    make, opt, want, order = families()[fam]
    opt = dict(opt)
    _lib.require_gpu()
    A, B = pair(make)
    # (mxLoop, tolerance): whole passes + a tail; one sweep; a stop the rule decides inside a pass
    runs = [(10, 0.0), (0, 0.0), (200, None)]
    ref = {}
    for name, q in (('A', A), ('B', B)):
        _, f = util.run_oracle(q, 12, 0.0, order)
        hist = []
        for n in (5, 6):
            _, g = util.run_oracle(q, n, 0.0, order)
            hist.append(g[1])
        tol_mid = 0.5 * (hist[0] + hist[1])              # stops at loop 6 when the change falls monotonically
        for mx, tol in runs:
            t = tol_mid if tol is None else tol
            ref[(name, mx)] = (t,) + util.run_oracle(q, mx, t, order)
    bad = 0
    shared = tuple(A.get('shared', ()))
    for it in range(nalt):
        for name, q in (('A', A), ('B', B)):
            mx = runs[it % len(runs)][0]
            tol, So, flo = ref[(name, mx)]
            S, fl, st = util.run_hip_batched([q], mx, tol, shared=shared, **opt)
            for k, v in want.items():
                if k == 'masked_min':
                    assert st['masked_tile_pct'] >= v, 'family %s: no tile was skipped: %r' % (fam, st)
                elif not (k == 'sweeps_per_launch' and mx == 0 and fam.startswith('fused3d')):   # (a one-sweep solve has no two-sweep pass)
                    assert st[k] == v, 'family %s ran %r, wanted %s = %r' % (fam, st, k, v)
            if not np.array_equal(S[0], So) or fl[0][2] != flo[2] or fl[0][0] != flo[0]:
                bad += 1
                print('MISMATCH family %s alternation %d problem %s mxLoop %d: %d points differ, loop %r vs %r'
                      % (fam, it, name, mx, int((S[0] != So).sum()), fl[0][2], flo[2]), flush=True)
    print('%s: %d alternations x 2 problems, %d mismatches' % (fam, nalt, bad))
    return 1 if bad else 0


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--list':
        print(' '.join(families()))
        sys.exit(0)
    sys.exit(main())
