"""Adversarial stop-rule cases (tests/stop_rule_edge.py): tolerances within one ulp of a sweep's relative change on the
full-size C2 and C5 problems, on every norm path -- lagged reducer, in-kernel reducer (XINV_LAG=0, read once per
process: hence the subprocesses), watchdog recovery, colour launches."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize('lag', ['1', '0'])
@pytest.mark.parametrize('cfg', ['c2', 'c5'])
def test_tolerance_within_an_ulp_of_a_sweeps_change(cfg, lag):
    from xinvert_amd import build as xbuild                  # (the watchdog-recovery path needs the test-hooks library)
    assert os.path.exists(xbuild.HOOKS_SO), 'build/libxinv_hooks.so is missing: python -m xinvert_amd.build --hooks'
    env = dict(os.environ); env['XINV_LAG'] = lag; env['XINV_SO'] = os.path.abspath(xbuild.HOOKS_SO)
    out = subprocess.run([sys.executable, os.path.join(HERE, 'stop_rule_edge.py'), cfg], capture_output=True, text=True,
                         timeout=1500, env=env)
    assert out.returncode == 0 and ': 0 wrong' in out.stdout, (out.stdout[-4000:], out.stderr[-3000:])
