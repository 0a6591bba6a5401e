"""Build the HIP extension in-tree:  python -m xinvert_amd.build

hipcc cross-compiles gfx950 code objects without a GPU; the resulting
xinvert_amd/libxinv_hip.so travels to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
SO = os.path.join(HERE, 'libxinv_hip.so')
SOURCES = ['xinv_hip.hip']
HEADERS = ['xinv_device.h', 'xinv_colour.h', 'xinv_fused.h', 'xinv_fused3d.h', 'xinv_fused3dg.h', 'xinv_fused9.h', 'xinv_fusedbih.h',
           'xinv_host.h', 'xinv_launch.h']
# -ffp-contract=off: no FMA contraction, so device results are bitwise those of the
# CPU restatement of the same sweep ordering (see DESIGN.md "Arithmetic").
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared']


def hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found')


def stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    deps.append(os.path.join(HERE, '..', 'include', 'xinv.h'))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not stale():
        return SO
    cmd = [hipcc()] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ['-o', SO]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return SO


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
