"""Tile ids and the seam launches' dispatch order (xinvert_amd/csrc/xinv_tiles.h): the header's integer arithmetic is shared
by the kernels and the planner; here it is compiled with g++ and checked exhaustively on small geometries (CPU)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which('g++') is None, reason='no g++')
def test_tile_rows_partition_and_heavy_first_is_a_bijection(tmp_path):
    exe = str(tmp_path / 'tiles_check')
    subprocess.run(['g++', '-O1', '-std=c++17', '-I', os.path.join(ROOT, 'xinvert_amd', 'csrc'),
                    os.path.join(ROOT, 'tests', 'csrc', 'tiles_check.cpp'), '-o', exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith('OK'), r.stdout + r.stderr
