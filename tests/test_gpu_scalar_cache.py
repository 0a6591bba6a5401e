"""Reuse stress of every kernel family: two different problems alternate through the same
workspace and device addresses, each family in a fresh process so that the first launch of its kernel
variants happens under the test; every solve bit for bit the oracle's, with the oracle's loop index.
See tests/stress_scalar_cache.py; the static side of the same rule is tests/test_smem_audit.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
FAMILIES = ['pipe2d', 'pipe2d_ext', 'pipe2d_skip', 'pipe2d_fr', 'pipe2d_gen', 'pipe2d_gen_fr', 'fused2d_std_um3', 'fused2d_std_um3_skip', 'fused2d_std_full',
            'fused2d_gen_um31', 'fused2d_gen_um28', 'fused2d_gen_full', 'fused2d_std2dt_um7', 'fused9_std',
            'fused9_gen', 'fused3d_uni', 'fused3d_full', 'fused3d_two_sweeps', 'fused3dg', 'fusedbih',
            'fusedbih_ext_per', 'fusedbih_vm1', 'fusedbih_vm2', 'bih_rowclass_uni', 'bih_colour_uni', 'colour_std2d_ext', 'colour_gen2d_nine',
            'colour_std3d_ext', 'fused2d_seam', 'fused2d_seam_um3', 'fused3d_seam', 'fused3d_seam_uni', 'fused3dg_seam', 'fused9_seam', 'fused3d_seam_ring', 'pipe2d_seam']


def test_family_list_is_complete():
    out = subprocess.run([sys.executable, os.path.join(HERE, 'stress_scalar_cache.py'), '--list'],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert sorted(out.stdout.split()) == sorted(FAMILIES)


@pytest.mark.parametrize('family', FAMILIES)
def test_alternating_problems_reuse_every_address(family):
    out = subprocess.run([sys.executable, os.path.join(HERE, 'stress_scalar_cache.py'), family, '50'],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-3000:])
    assert '0 mismatches' in out.stdout, out.stdout[-2000:]
