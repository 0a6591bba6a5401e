"""The watchdog-recovery suite (tests/hooks_suite) needs the test-hooks variant of the library -- a tile that withholds
its norm partial, a 30 ms watchdog, the XINV_EXP_WATCHDOG switch: none of it is in the shipped libxinv_hip.so -- and a
process binds ONE library (XINV_SO is read at import), so the suite runs in a child process against
build/libxinv_hooks.so (xinvert_amd.build.build_hooks; built by __graft_entry__.build())."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_shipped_library_has_no_test_hooks():
    """The environment switches do nothing on the shipped library: nothing is recovered, results are the oracle's."""
    import numpy as np
    import util
    from oracle import COLOUR_2
    p = util.rand2d('std2d', 61, 150, 'fixed', 'periodic', msk=True, seed=1)
    So, flo = util.run_oracle(p, 21, 0.0, COLOUR_2)
    os.environ['XINV_EXP_WATCHDOG'] = '1,0'
    os.environ['XINV_HOOK_SKIP_PUBLISH'] = '1,0,0'
    try:
        S, fl, st = util.run_hip_dev([p], 21, 0.0)
    finally:
        os.environ.pop('XINV_EXP_WATCHDOG', None); os.environ.pop('XINV_HOOK_SKIP_PUBLISH', None)
    assert st['recovered_members'] == 0 and np.array_equal(S[0], So) and fl[0][2] == flo[2]


def test_shipped_library_reads_no_tuning_switch_from_the_environment():
    """a stray XINV_* variable in a user's shell must not alter a production solve.  The planner's overrides
    are fields of xinv_options (lanes, norm_lag, pipe_fr, graph); the shipped library has no getenv at all."""
    import numpy as np
    import util
    ps = [util.rand2d('std2d', 700, 1500, 'fixed', 'periodic', msk=False, seed=s) for s in (1, 2, 3, 4)]
    for q in ps:                                          # A and C constant along x: the pipelined pass, four sweeps per launch
        for k in (0, 2):
            q['coefs'][k] = np.repeat(q['coefs'][k][:, :1], q['coefs'][k].shape[1], axis=1)
    keys = dict(XINV_LANES='1', XINV_LAG='0', XINV_PIPE='0', XINV_PIPE_FR='1', XINV_3D_K2='0', XINV_GRAPH='1',
                XINV_EXP_NOCTL='1', XINV_PIPE_OCC='1', XINV_PIN='1')
    S0, f0, st0 = util.run_hip_dev(ps, 12, 0.0)
    os.environ.update(keys)
    try:
        S1, f1, st1 = util.run_hip_dev(ps, 12, 0.0)
    finally:
        for k in keys:
            os.environ.pop(k, None)
    assert np.array_equal(S0, S1) and np.array_equal(f0, f1)
    for k in ('lanes', 'pipelined', 'sweeps_per_launch', 'rows_per_tile', 'path', 'sweep_launches'):
        assert st0[k] == st1[k], (k, st0[k], st1[k])
    assert st0['lanes'] == 2, st0                         # (the rule cut this batch in two chains, XINV_LANES=1 notwithstanding)
    # (statically, in the CPU suite: tests/test_host.py::test_shipped_library_does_not_import_getenv)


def test_watchdog_recovery_suite_on_the_hooks_library():
    from xinvert_amd import build as xbuild
    so = xbuild.HOOKS_SO
    assert os.path.exists(so), 'build/libxinv_hooks.so is missing: python -m xinvert_amd.build --hooks'
    #  ... and linked from objects compiled from the CURRENT sources: a stale variant would run old kernels
    assert xbuild.fresh(tag=xbuild.HOOKS_TAG, extra=['-DXINV_TEST_HOOKS=1'], variant_units=xbuild.HOOKS_UNITS), \
        'build/libxinv_hooks.so is stale: python -m xinvert_amd.build --hooks'
    env = dict(os.environ); env.update(XINV_SO=os.path.abspath(so), XINV_HOOKS_SUITE='1')
    out = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu', os.path.join(HERE, 'hooks_suite')],
                         capture_output=True, text=True, timeout=2400, env=env, cwd=os.path.dirname(HERE))
    tail = out.stdout[-3000:]
    assert out.returncode == 0 and ' passed' in tail and 'failed' not in tail, (tail, out.stderr[-2000:])
