"""Resident plans at the C-ABI (include/xinv.h, xinv_plan_*): everything a solve derives from the coefficient stack is
built ONCE and reused -- the reference calls its kernel again and again on one stack (apps.animate_iteration,
apps.py:1031-1044; tests/test_AnimateConverge.py:13-31).  A solve on a plan must be bit for bit the solve of the same
arrays through xinv_<form>_f64_dev -- and the oracle's -- whatever happens on the device between two solves."""
import ctypes

import numpy as np
import pytest

import util
from oracle import COLOUR_2, COLOUR_AUTO

pytestmark = pytest.mark.gpu


def _mk(kind, msk, seed, big=False):
    if kind == 'std3d':
        return util.rand3d(14, 40, 300, 'fixed', 'periodic', msk, seed=seed)
    if kind == 'gen3d':
        return util.rand3dg(12, 36, 280, 'fixed', 'periodic', msk, seed=seed)
    if kind == 'bih2d':
        return util.randbih(60, 360, 'fixed', 'fixed', 0, msk, seed=seed)
    if kind == 'std2dt':
        return util.rand2dt(70, 300, 'extend', 'periodic', 0, msk, seed=seed)
    yc, xc = (400, 1200) if big else (90, 420)
    return util.rand2d(kind, yc, xc, 'fixed', 'periodic', 0, msk, seed=seed)


def _uniform(p):
    """Every coefficient array but the forcing constant along x (what the lat-lon builders produce)."""
    q = dict(p)
    q['coefs'] = [np.repeat(c[..., :1], c.shape[-1], axis=-1) for c in p['coefs'][:-1]] + [np.array(p['coefs'][-1], copy=True)]
    return q


def _as_problem(ps, shared=(), rowviews=False):
    """The dict ResidentProblem takes: stacked members; arrays in `shared` given once; rowviews: x-uniform arrays as
    stride-0 views along x (uploaded as ONE value per row: xinv_options.rowconst_mask)."""
    p = dict(ps[0])
    p['S0'] = np.stack([q['S0'] for q in ps])
    cs = []
    for k in range(len(ps[0]['coefs'])):
        a = ps[0]['coefs'][k] if k in shared else np.stack([q['coefs'][k] for q in ps])
        if rowviews and k < len(ps[0]['coefs']) - 1:
            a = np.broadcast_to(np.ascontiguousarray(a[..., :1]), a.shape)
        cs.append(a)
    p['coefs'] = cs
    p['shared'] = tuple(shared)
    return p


@pytest.mark.parametrize('kind', ['std2d', 'gen2d', 'std2dt', 'bih2d', 'std3d', 'gen3d'])
@pytest.mark.parametrize('uni', [0, 1, 2])
def test_plan_solve_equals_dev_entry_and_oracle(kind, uni):
    """uni = 0: every coefficient array varies along x; 1: constant along x, handed over as full arrays (the plan finds
    out once); 2: handed over as one value per row (rowconst_mask: expanded by the plan, never tested)."""
    from xinvert_amd.resident import ResidentProblem
    ps = [_mk(kind, m & 1, 100 + 7 * m) for m in range(3)]
    if uni:
        ps = [_uniform(q) for q in ps]
    ncoef = len(ps[0]['coefs'])
    shared = tuple(range(ncoef - 1)) if uni else ()
    if uni:
        for q in ps[1:]:
            q['coefs'][:ncoef - 1] = ps[0]['coefs'][:ncoef - 1]
    order = COLOUR_AUTO
    mx, tol = 17, 0.0
    ref = [util.run_oracle(q, mx, tol, order) for q in ps]
    prob = _as_problem(ps, shared, rowviews=(uni == 2))
    Sd, fd, std = util.run_hip_dev(ps, mx, tol, shared=shared)
    rp = ResidentProblem(prob, plan=True, null_zero_B=True)
    assert (rp.rowconst != 0) == (uni == 2)
    for rep in range(3):                                  # the plan is built by the first solve and reused by the others
        rp.reset()
        fl, st = rp.solve(mx, tol)
        S = rp.result()
        assert st['planned'] == 1 and st['plan_ms'] == 0.0, st
        for k in ('path', 'sweeps_per_launch', 'xuniform_mask', 'pipelined', 'colours'):
            assert st[k] == std[k], (k, st[k], std[k])
        assert np.array_equal(S, Sd, equal_nan=True) and np.array_equal(fl, fd, equal_nan=True), (kind, uni, rep)
        for m in range(3):
            assert np.array_equal(S[m], ref[m][0], equal_nan=True), (kind, uni, rep, m)
            assert fl[m][2] == ref[m][1][2]
    assert len(rp._plans) == 1
    rp.close()


@pytest.mark.parametrize('kind', ['std2d', 'gen2d'])
def test_plan_restarts_continue_like_the_reference_kernels(kind):
    """animate_iteration's pattern (apps.py:1031-1044): frame after frame of a few sweeps on the same S, in place; with
    masked tiles skipped (lists kept by the plan) and tolerance stops."""
    from xinvert_amd.resident import ResidentProblem
    p = _uniform(_mk(kind, 1, 5, big=True))
    F = p['coefs'][-1]
    F[40:260, 100:700] = p['undef']                       # whole tiles to skip
    prob = _as_problem([p], shared=tuple(range(len(p['coefs']) - 1)), rowviews=True)
    rp = ResidentProblem(prob)
    q = dict(p)
    for frame in range(5):
        fl, st = rp.solve(7, 0.0, force_tile_skip=1)      # eight sweeps per frame: two full passes on the tile lists
        assert st['planned'] == 1 and st['masked_tile_pct'] > 0, st
        So, flo = util.run_oracle(q, 7, 0.0, COLOUR_2)
        assert np.array_equal(rp.result()[0], So), frame
        q = dict(q); q['S0'] = So
    fl, st = rp.solve(400, 1e-4, force_tile_skip=1)       # ... then to a tolerance, stopping inside a pass
    So, flo = util.run_oracle(q, 400, 1e-4, COLOUR_2)
    assert fl[0][2] == flo[2] and np.array_equal(rp.result()[0], So)
    rp.close()


def test_plans_survive_other_solves_on_the_device():
    """Two plans of different geometry, used alternately, with plain *_dev and host-pointer solves of a third problem in
    between: a plan's records and tile lists are its own buffers (swapped into the workspace per solve), not the
    workspace's."""
    from xinvert_amd.resident import ResidentProblem
    pa = _uniform(_mk('std2d', 1, 11, big=True)); pa['coefs'][-1][10:200, 300:900] = pa['undef']
    pb = _uniform(_mk('gen2d', 0, 12))
    pc = util.rand3d(12, 60, 260, 'fixed', 'periodic', 1, seed=13)
    pd = _uniform(util.rand2d('std2d', 700, 1500, 'fixed', 'fixed', 0, 1, seed=14))     # (larger records and lists than A's)
    ra = ResidentProblem(_as_problem([pa], (0, 1, 2), rowviews=True))
    rb = ResidentProblem(_as_problem([pb], (), rowviews=False))
    refa = util.run_oracle(pa, 13, 0.0, COLOUR_2)[0]
    refb = util.run_oracle(pb, 9, 0.0, COLOUR_2)[0]
    for it in range(3):
        ra.reset(); ra.solve(13, 0.0, force_tile_skip=1)
        assert np.array_equal(ra.result()[0], refa), it
        util.run_hip_dev([pd], 5, 0.0, force_tile_skip=1)
        rb.reset(); rb.solve(9, 0.0)
        assert np.array_equal(rb.result()[0], refb), it
        util.run_hip_batched([pc], 4, 0.0)
        util.run_hip_dev([pd, pd], 6, 0.0)
    ra.close(); rb.close()


def test_plan_refresh_after_the_coefficients_changed():
    import torch
    from xinvert_amd.resident import ResidentProblem
    p = _mk('std2d', 1, 21)
    rp = ResidentProblem(_as_problem([p], ()))
    rp.solve(8, 0.0)
    assert np.array_equal(rp.result()[0], util.run_oracle(p, 8, 0.0, COLOUR_2)[0])
    # new coefficients, constant along x now, and another mask: everything the plan holds is stale
    q = _uniform(_mk('std2d', 0, 22))
    q['coefs'][-1][5:40, 50:300] = q['undef']
    for k in (0, 2, 3):
        rp.coefs[k].copy_(torch.from_numpy(np.ascontiguousarray(q['coefs'][k])[None]).to(rp.dev))
    rp.S0.copy_(torch.from_numpy(q['S0'][None]).to(rp.dev)); rp.reset()
    rp.refresh()
    fl, st = rp.solve(8, 0.0)
    assert st['xuniform_mask'] == 3 and st['planned'] == 1, st
    assert np.array_equal(rp.result()[0], util.run_oracle(q, 8, 0.0, COLOUR_2)[0])
    rp.close()


def test_plan_solves_complete_in_stream_order_on_any_stream():
    """xinv_plan_solve_f64_dev returns with the flags final and S complete IN STREAM ORDER (include/xinv.h): two problems
    solved alternately on two non-default streams -- every solve leaves a copy of its final state queued behind it, the next
    solve, on the other stream, reuses the workspace's ping-pong buffers -- and read back on the stream they were solved on:
    bit for bit the oracle, short solves (one launch) and tolerance stops inside a pass."""
    import torch
    from xinvert_amd.resident import ResidentProblem
    ps = [_uniform(_mk('std2d', 0, 51 + k)) for k in range(2)]
    rps = [ResidentProblem(_as_problem([q], (0, 1, 2))) for q in ps]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for mx, tol in ((2, 0.0), (13, 0.0), (40, 3e-3), (1, 0.0)):
        refs = [util.run_oracle(q, mx, tol, COLOUR_2) for q in ps]
        for rep in range(6):
            outs = []
            for k in (0, 1):
                with torch.cuda.stream(streams[k]):
                    rps[k].reset()
                    fl, st = rps[k].solve(mx, tol)
                    outs.append((rps[k].S.clone(), np.array(fl, copy=True), st))       # (the clone is queued on the same stream)
            for k in (0, 1):
                streams[k].synchronize()
                S, fl, st = outs[k]
                assert st['planned'] == 1 and fl[0][2] == refs[k][1][2], (fl, refs[k][1])
                assert np.array_equal(S.cpu().numpy()[0], refs[k][0]), 'problem %d mxLoop %d rep %d' % (k, mx, rep)
    for rp in rps:
        rp.close()


def test_plan_argument_errors():
    import torch
    from xinvert_amd import _lib
    from xinvert_amd.resident import ResidentProblem
    L = _lib.require_gpu()
    p = _uniform(_mk('std2d', 0, 31))
    rp = ResidentProblem(_as_problem([p], (0, 1, 2)))
    rp.solve(3, 0.0)
    h = next(iter(rp._plans.values()))
    st = torch.cuda.current_stream()
    # S off the 16-byte grid: the plan's kernels use 16-byte accesses
    big = torch.zeros(rp.n + 2, dtype=torch.float64, device=rp.dev)
    rc = L.xinv_plan_solve_f64_dev(h, ctypes.c_void_p(big.data_ptr() + 8), _lib.hptr(rp.flags), 3, 0.0,
                                   ctypes.c_void_p(st.cuda_stream))
    assert rc == -1 and b'16-byte' in L.xinv_last_error()
    assert L.xinv_plan_solve_f64_dev(None, ctypes.c_void_p(big.data_ptr()), _lib.hptr(rp.flags), 3, 0.0, None) == -1
    assert L.xinv_plan_solve_f64_dev(h, None, _lib.hptr(rp.flags), 3, 0.0, None) == -1
    assert L.xinv_plan_solve_f64_dev(h, ctypes.c_void_p(rp.S.data_ptr()), _lib.hptr(rp.flags), -1, 0.0, None) == -1
    assert L.xinv_plan_destroy(None) == 0
    # a row-constant forcing is refused
    o = _lib.options(rowconst_mask=8)
    hh = ctypes.c_void_p()
    from xinvert_amd.resident import PLAN_FN, scalars
    ptr = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    rc = getattr(L, PLAN_FN['std2d'])(ctypes.byref(hh), *[ptr(c) for c in rp.coefs], rp.nb, rp.strides, *scalars(rp.p),
                                      ctypes.byref(o), None)
    assert rc == -1 and not hh.value
    rp.close()


def test_animate_iteration_runs_on_a_plan_and_equals_the_host_path():
    """apps.animate_iteration (reference apps.py:895-1058; tests/test_AnimateConverge.py:13-31: Gill-Matsuno, 73x144):
    the frames of the resident path -- one plan, per-row coefficients as rows -- against the frames of the host-pointer
    inv_* call per frame."""
    import xinvert_amd as xa
    from xinvert_amd import apps, core
    lat = np.linspace(-90, 90, 73); lon = np.linspace(0, 360, 144, endpoint=False)
    Q = 0.05 * np.exp(-((lat[:, None] - 0.0) ** 2 + (lon[None, :] - 120.0) ** 2) / 100.0)
    F = xa.Field(Q, ('lat', 'lon'), {'lat': lat, 'lon': lon})
    kw = dict(dims=['lat', 'lon'], coords='lat-lon', mParams={'epsilon': 1e-5, 'Phi': 5000.0},
              iParams={'BCs': ['fixed', 'periodic'], 'tolerance': 1e-12, 'optArg': 1.4}, loop_per_frame=2, max_frames=12)
    ip1 = dict(kw['iParams'])
    a = apps.animate_iteration('GillMatsuno', F, **dict(kw, iParams=ip1))
    assert ip1['stats']['planned'] == 1, ip1['stats']
    saved = core.Resident
    try:
        core.Resident = None                              # (the frames through the host-pointer call, as the reference)

        def _raise(*a_, **k_):
            raise ImportError
        core.Resident = _raise
        ip2 = dict(kw['iParams'])
        b = apps.animate_iteration('GillMatsuno', F, **dict(kw, iParams=ip2))
    finally:
        core.Resident = saved
    assert ip2['stats']['planned'] == 0
    assert np.array_equal(np.asarray(a.values), np.asarray(b.values))
    assert np.array_equal(ip1['frame_flags'], ip2['frame_flags'])


def test_float32_forcing_with_an_undef_float32_cannot_hold_exactly():
    """a float32 forcing filled with 9.96921e36 (netCDF's default fill) or 1e20 carries float32(undef); the
    device-side mask pass compares the promoted value with THAT, as the reference's `F.where(F != undef)` does in float32
    (apps.py:2124-2128) -- the fill points are masked, not solved as data."""
    import xinvert_amd as xa
    rng = np.random.default_rng(3)
    lat = np.linspace(-89.5, 89.5, 90); lon = np.arange(0.0, 360.0, 2.0)
    z = (1e-5 * rng.standard_normal((2, 90, 180))).astype(np.float32)
    land = rng.random((90, 180)) < 0.2
    for undef in (9.96921e36, 1e20, -9999.0):
        z32 = z.copy(); z32[:, land] = np.float32(undef)
        z64 = z.astype(np.float64); z64[:, land] = np.nan
        out = {}
        for tag, arr, ud in (('f32', z32, undef), ('f64nan', z64, np.nan)):
            ip = {'BCs': ['fixed', 'periodic'], 'mxLoop': 60, 'tolerance': 1e-14, 'undef': ud, 'printInfo': False}
            F = xa.Field(arr, ('time', 'lat', 'lon'), {'lat': lat, 'lon': lon})
            out[tag] = np.asarray(xa.invert_Poisson(F, dims=['lat', 'lon'], coords='lat-lon', iParams=ip).values, dtype=np.float64)
        sea = ~land
        assert np.array_equal(out['f32'][:, sea], out['f64nan'][:, sea]), undef
        # the host fallback (device_prep off) masks the same points
        ip = {'BCs': ['fixed', 'periodic'], 'mxLoop': 60, 'tolerance': 1e-14, 'undef': undef, 'printInfo': False,
              'device_prep': False}
        F = xa.Field(z32, ('time', 'lat', 'lon'), {'lat': lat, 'lon': lon})
        host = np.asarray(xa.invert_Poisson(F, dims=['lat', 'lon'], coords='lat-lon', iParams=ip).values, dtype=np.float64)
        assert np.array_equal(host[:, sea], out['f64nan'][:, sea]), undef


@pytest.mark.parametrize('mode', ['', '--stops'])
def test_plan_fuzz(mode):
    """tests/fuzz_plan.py: the medium seeded generator (every form, masks, 'extend', odd widths, batches of two) with
    every case solved on a plan, re-solved on the same plan and continued from its result -- against the oracle."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, 'fuzz_plan.py'), '2000', '6'] + ([mode] if mode else []),
                         capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0 and 'failures: 0' in out.stdout, (out.stdout[-3000:], out.stderr[-3000:])


@pytest.mark.parametrize('kind', ['std2d', 'gen2d', 'std2dt', 'bih2d', 'std3d', 'gen3d'])
@pytest.mark.parametrize('mxLoop,tol,nf', [(1, 0.0, 9), (4, 0.0, 5), (2, 3e-3, 40)])
def test_plan_solve_frames_equals_repeated_solves(kind, mxLoop, tol, nf):
    """xinv_plan_solve_frames_f64_dev: nf restarts queued behind each other (apps.animate_iteration's frames) are, bit for bit,
    nf calls of xinv_plan_solve_f64_dev -- the snapshot of every frame, the flags of every frame, the final S.  Budgets of
    one launch and of several (with a shorter tail launch), a masked forcing (skipped tiles), and a tolerance that stops some
    frame early INSIDE the sequence: the fast road's result up to there, the ordinary road from there on."""
    import torch
    from xinvert_amd.resident import ResidentProblem
    ps = [_mk(kind, 1, 70 + 3 * m) for m in range(2)]
    p = _as_problem(ps)
    ref = ResidentProblem(p, null_zero_B=True)
    ref_frames, ref_flags = [], []
    for f in range(nf):
        fl, st = ref.solve(mxLoop, tol)
        ref_frames.append(ref.S.clone()); ref_flags.append(np.array(fl, copy=True))
    rp = ResidentProblem(p, null_zero_B=True)
    frames = torch.empty((nf,) + tuple(rp.S.shape), dtype=rp.S.dtype, device=rp.S.device)
    fl, st = rp.solve_frames(frames, mxLoop, tol)
    assert fl.shape == (nf, rp.nb, 3)
    for f in range(nf):
        assert torch.equal(frames[f], ref_frames[f]), 'frame %d differs' % f
        assert np.array_equal(fl[f][:, [0, 2]], ref_flags[f][:, [0, 2]]), (f, fl[f], ref_flags[f])
        assert np.allclose(fl[f][:, 1], ref_flags[f][:, 1], rtol=1e-12, atol=0, equal_nan=True)
    assert torch.equal(rp.S, ref.S)
    if tol > 0:                                          # (the sequence must really have hit an early stop)
        assert any((ref_flags[f][:, 2] < mxLoop).any() for f in range(nf)), [r[:, 2] for r in ref_flags]
    # and a second call continues from there
    fl2, _ = rp.solve_frames(frames[:2], mxLoop, tol)
    r1, _ = ref.solve(mxLoop, tol)
    assert torch.equal(frames[0], ref.S) and np.array_equal(fl2[0][:, [0, 2]], r1[:, [0, 2]])
