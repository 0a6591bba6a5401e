"""The sweep loop in lanes (run_sweeps): the batch cut into independent launch chains on separate
streams must give every member bitwise the single-chain result -- the oracle's -- whatever the number of lanes, with
members stopping at different sweeps.  One child process per setting (xinv_options.lanes; 'auto' = the engine's rule)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize('lanes', ['auto', '1', '2', '3', '4'])
def test_lanes_give_the_oracles_result(lanes):
    env = dict(os.environ)
    out = subprocess.run([sys.executable, os.path.join(HERE, 'lanes_case.py'), '0' if lanes == 'auto' else lanes],
                         capture_output=True, text=True, timeout=1500, env=env)
    assert out.returncode == 0 and 'lanes ok' in out.stdout, (out.stdout[-3000:], out.stderr[-3000:])


@pytest.mark.parametrize('lanes', ['auto', '2', '3'])
def test_lanes_fuzz(lanes):
    """tests/fuzz_lanes.py: 40 seeded batches of 2..24 members of every form, members stopping on the tolerance at
    different sweeps, every member against the oracle."""
    env = dict(os.environ)
    out = subprocess.run([sys.executable, os.path.join(HERE, 'fuzz_lanes.py'), '1000', '40'] + ([] if lanes == 'auto' else ['--lanes=' + lanes]),
                         capture_output=True, text=True, timeout=1500, env=env)
    assert out.returncode == 0 and 'failures: 0' in out.stdout, (out.stdout[-3000:], out.stderr[-3000:])


def test_single_member_fuzz_with_skipped_tiles():
    """tests/fuzz_lanes.py --single: one member, blocks of the forcing blanked (masked-tile lists), grids large enough for
    the lagged norm -- three buffers, the skipped tiles' norm share and copies on the side stream beside the first launch,
    tolerance stops -- against the oracle."""
    env = dict(os.environ)
    out = subprocess.run([sys.executable, os.path.join(HERE, 'fuzz_lanes.py'), '60000', '40', '--single'],
                         capture_output=True, text=True, timeout=1500, env=env)
    assert out.returncode == 0 and 'failures: 0' in out.stdout and ' 0 with skipped tiles' not in out.stdout, \
        (out.stdout[-3000:], out.stderr[-3000:])
