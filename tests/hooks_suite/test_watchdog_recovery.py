"""Watchdog recovery (xinv_hip.hip, run_sweeps): a member whose in-kernel norm reduction gave up waiting for a
partial is finished sweep by sweep -- one-sweep launches without in-kernel norm + the separate norm kernels --
instead of failing the call.  Runs against build/libxinv_hooks.so (the test-hooks variant; tests/test_gpu_watchdog.py
starts this suite in its own process with XINV_SO set): the shipped library has no hook.  Two hooks:
  XINV_HOOK_SKIP_PUBLISH="launch,tile[,member]"  one tile of one launch WITHHOLDS its norm partial: the reducer of that
      launch -- the launch's last workgroup, or with the lagged norm the extra workgroup of the NEXT launch, while that
      launch's tiles write the third buffer -- really runs into its watchdog (30 ms in this build);
  XINV_EXP_WATCHDOG="launch[,member]" puts the member's control block, before that launch, into exactly the state a
      timed-out reducer leaves behind (stopped with overflow = 2, the stop rule applied to no sweep of the launch).
Everything after that is the production code: finding the launch boundary, the intact source buffer (two buffers, or
three with the lagged norm), the resumed sweeps, the stop rule, where the final state lies.  Results must equal the
oracle's coloured ordering bit for bit, the loop index exactly, flags[1] to rounding (the separate kernels add the
partial sums in another order); the other members of the batch must not notice."""
import os

import numpy as np
import pytest

import util
from util import run_oracle
from oracle import COLOUR_2, COLOUR_AUTO

pytestmark = pytest.mark.gpu


def test_this_is_the_hooks_library():
    from xinvert_amd import _lib
    assert 'libxinv_hooks' in _lib.SO, 'run through tests/test_gpu_watchdog.py (XINV_SO=build/libxinv_hooks.so)'


def _cases():
    from xinvert_amd import synthetic
    c = {}
    # small random problems: k_fused2d K = 2..4 / k_fused3d / k_pipe3d / k_fusedbih, in-kernel reducer, two buffers
    c['std2d'] = (lambda: [util.rand2d('std2d', 61, 150, 'fixed', 'periodic', msk=True, seed=s) for s in (1, 2, 3)], (), COLOUR_2)
    c['gen2d'] = (lambda: [util.rand2d('gen2d', 40, 130, 'extend', 'fixed', msk=False, seed=s) for s in (4, 5)], (), COLOUR_2)
    c['std3d'] = (lambda: [util.rand3d(9, 30, 130, 'fixed', 'periodic', msk=True, seed=s) for s in (6, 7)], (), COLOUR_2)

    def uni3d():
        ps = [util.rand3d(12, 40, 256, 'fixed', 'periodic', msk=True, seed=s) for s in (8, 9)]
        for p in ps:
            for q in range(3):
                p['coefs'][q][:] = p['coefs'][q][:, :, :1]
        return ps
    c['std3d_two_sweeps'] = (uni3d, (), COLOUR_2)

    def bih():
        ps = [util.randbih(48, 200, 'fixed', 'fixed', msk=False, seed=s) for s in (10, 11)]
        for p in ps:
            for q in range(9):
                p['coefs'][q][:] = p['coefs'][q][:, :1]
        return ps
    c['bih2d'] = (bih, (), COLOUR_AUTO)

    # lat-lon Poisson, large enough for the lagged norm (three buffers) on the pipelined pass, with tile skipping
    def latlon():
        p = synthetic.poisson_latlon(360, 720, mask=True, members=2)
        return [synthetic.member(p, m) for m in range(2)], p['shared']
    c['latlon_pipelined_lagged'] = (latlon, None, COLOUR_2)
    return c


@pytest.mark.parametrize('at', [0, 1, 3])
@pytest.mark.parametrize('stop', ['budget', 'tolerance'])
@pytest.mark.parametrize('name', ['std2d', 'gen2d', 'std3d', 'std3d_two_sweeps', 'bih2d', 'latlon_pipelined_lagged'])
def test_watchdog_recovery_equals_oracle(name, stop, at):
    make, shared, order = _cases()[name]
    ps = make()
    if shared is None:
        ps, shared = ps
    nm = len(ps)
    victim = nm - 1
    mx, tol = 21, 0.0
    if stop == 'tolerance':                               # a tolerance the victim meets after ten or so sweeps,
        pre = run_oracle(ps[victim], 13, 0.0, order)[1]   # well clear of the values around it
        mx, tol = 400, 1.5 * pre[1]
    ref = [run_oracle(q, mx, tol, order) for q in ps]
    if stop == 'tolerance':
        assert at < ref[victim][1][2] < 14, 'the case should stop on tolerance after the injected launch: %r' % (ref[victim][1],)
    os.environ['XINV_EXP_WATCHDOG'] = '%d,%d' % (at, victim)
    try:
        S, fl, st = util.run_hip_dev(ps, mx, tol, shared=shared)
    finally:
        os.environ.pop('XINV_EXP_WATCHDOG', None)
    assert st['recovered_members'] == 1, st
    if name == 'latlon_pipelined_lagged':
        assert st['pipelined'] == 1, st
    for m in range(nm):
        So, flo = ref[m]
        assert np.array_equal(S[m], So), '%s member %d: %d points differ (%r)' % (name, m, int((S[m] != So).sum()), st)
        assert fl[m][2] == flo[2], (m, fl[m], flo)
        assert abs(fl[m][1] - flo[1]) <= 1e-11 * max(1.0, abs(flo[1])), (m, fl[m], flo)
        assert fl[m][0] == flo[0]
    # and without the hook nothing is recovered
    S2, fl2, st2 = util.run_hip_dev(ps, mx, tol, shared=shared)
    assert st2['recovered_members'] == 0 and np.array_equal(S2, S)


# ------------------------------------------------------------------ a REAL reducer timeout
@pytest.mark.parametrize('lag', ['1', '0'])
@pytest.mark.parametrize('at,tile', [(0, 0), (1, 1), (2, 2)])
@pytest.mark.parametrize('name', ['std2d', 'gen2d', 'latlon_pipelined_lagged', 'latlon_fused2d'])
def test_real_reducer_timeout_is_recovered(name, at, tile, lag):
    """One tile of launch `at` withholds its partial (XINV_HOOK_SKIP_PUBLISH): the reducer of that launch times out
    for real -- in-kernel (XINV_LAG=0 is read once per process, so the lag = '0' cases run in a child process), or, with
    the lagged norm, in the extra workgroup of launch at+1 while that launch's tiles write the third buffer -- and the
    recovery must still find the launch boundary and an intact source buffer: bitwise the oracle, loop index exact."""
    import subprocess
    import sys
    if lag == '0':
        env = dict(os.environ); env.update(XINV_LAG='0', XINV_HOOKS_SUITE='1')
        out = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', __file__, '-k',
                              'test_real_reducer_timeout_is_recovered and %s-%d-%d-1' % (name, at, tile)],
                             capture_output=True, text=True, timeout=900, env=env)
        assert out.returncode == 0 and '1 passed' in out.stdout, (out.stdout[-3000:], out.stderr[-2000:])
        return
    make, shared, order = (_cases()[name] if name != 'latlon_fused2d' else _cases()['latlon_pipelined_lagged'])
    ps = make()
    if shared is None:
        ps, shared = ps
    opt = dict(no_pipe=1) if name == 'latlon_fused2d' else {}
    if not name.startswith('latlon'):
        tile = 0                                          # (the small cases have one or two workgroups per member)
    nm = len(ps)
    victim = nm - 1
    mx, tol = 21, 0.0
    ref = [run_oracle(q, mx, tol, order) for q in ps]
    os.environ['XINV_HOOK_SKIP_PUBLISH'] = '%d,%d,%d' % (at, tile, victim)
    try:
        S, fl, st = util.run_hip_dev(ps, mx, tol, shared=shared, **opt)
    finally:
        os.environ.pop('XINV_HOOK_SKIP_PUBLISH', None)
    assert st['recovered_members'] == 1, st
    for m in range(nm):
        So, flo = ref[m]
        assert np.array_equal(S[m], So), '%s member %d: %d points differ (%r)' % (name, m, int((S[m] != So).sum()), st)
        assert fl[m][2] == flo[2] and fl[m][0] == flo[0], (m, fl[m], flo)
        assert abs(fl[m][1] - flo[1]) <= 1e-11 * max(1.0, abs(flo[1])), (m, fl[m], flo)
    S2, fl2, st2 = util.run_hip_dev(ps, mx, tol, shared=shared, **opt)
    assert st2['recovered_members'] == 0 and np.array_equal(S2, S)
