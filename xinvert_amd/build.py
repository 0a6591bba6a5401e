"""Build the HIP extension in-tree:  python -m xinvert_amd.build [--force] [-jN]

hipcc cross-compiles gfx950 code objects without a GPU; the resulting
xinvert_amd/libxinv_hip.so travels to the GPU box with the repo snapshot.

The library is several translation units -- the host driver + C-ABI, and one unit per family of
templated sweep kernels -- compiled in parallel into build/obj/ and linked into one shared object.
Only units whose sources (or any header) changed are recompiled.
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
# A/B variants (tuning switches of the kernels): XINV_EXTRA_FLAGS="-DXINV_STAGE_SKIP=0"
# XINV_BUILD_TAG=noskip  ->  build/libxinv_noskip.so (objects in build/obj_noskip); run it with XINV_SO=...
TAG = os.environ.get('XINV_BUILD_TAG', '')
EXTRA = os.environ.get('XINV_EXTRA_FLAGS', '').split()
# (object name, source, extra flags)
UNITS = [
    ('xinv_hip', 'xinv_hip.hip', []),
    ('xinv_tu_fused2d_std', 'xinv_tu_fused2d.hip', ['-DXINV_TU_MODEL=0']),
    ('xinv_tu_fused2d_gen', 'xinv_tu_fused2d.hip', ['-DXINV_TU_MODEL=1']),
    ('xinv_tu_fused2d_std2dt', 'xinv_tu_fused2d.hip', ['-DXINV_TU_MODEL=2']),
    # the odd-xc periodic seam variants of the same kernels (unaligned strips only)
    ('xinv_tu_fused2d_std_seam', 'xinv_tu_fused2d.hip', ['-DXINV_TU_MODEL=0', '-DXINV_TU_SEAM=1']),
    ('xinv_tu_fused2d_gen_seam', 'xinv_tu_fused2d.hip', ['-DXINV_TU_MODEL=1', '-DXINV_TU_SEAM=1']),
    ('xinv_tu_fused2d_std2dt_seam', 'xinv_tu_fused2d.hip', ['-DXINV_TU_MODEL=2', '-DXINV_TU_SEAM=1']),
    ('xinv_tu_pipe2d_std', 'xinv_tu_pipe2d.hip', ['-DXINV_TU_MODEL=0']),
    ('xinv_tu_pipe2d_gen', 'xinv_tu_pipe2d.hip', ['-DXINV_TU_MODEL=1']),
    # contracted arithmetic (XINV_FLAG_FMA): the per-row-coefficient variants on the F models
    ('xinv_tu_fused2d_stdf', 'xinv_tu_fused2d.hip', ['-DXINV_TU_MODEL=3']),
    ('xinv_tu_fused2d_genf', 'xinv_tu_fused2d.hip', ['-DXINV_TU_MODEL=4']),
    # general form with coefficient arrays that vary along x: the point-factor stream (FusedGen2DQ, round 5)
    ('xinv_tu_fused2d_genq', 'xinv_tu_fused2d.hip', ['-DXINV_TU_MODEL=5']),
    ('xinv_tu_pipe2d_fma', 'xinv_tu_pipe2d.hip', ['-DXINV_TU_MODEL=2']),
    ('xinv_tu_pipe2d_std_seam', 'xinv_tu_pipe2d.hip', ['-DXINV_TU_MODEL=0', '-DXINV_TU_SEAM=1']),
    ('xinv_tu_pipe2d_gen_seam', 'xinv_tu_pipe2d.hip', ['-DXINV_TU_MODEL=1', '-DXINV_TU_SEAM=1']),
    ('xinv_tu_fused9', 'xinv_tu_fused9.hip', []),
    ('xinv_tu_fused3d', 'xinv_tu_fused3d.hip', []),
    ('xinv_tu_fused3d_fma', 'xinv_tu_fused3d_fma.hip', []),
    ('xinv_tu_fused3d_seam', 'xinv_tu_fused3d_seam.hip', []),
    # k_pipe3d: sixteen wavefronts x 128 VGPRs; the default scheduler's interleaving spills, minimum-register scheduling
    # fits the ring variant in 123 (C5, 15 volumes: 2.98 -> 3.26e11, profiles/r04_pipe3d_variants.txt)
    ('xinv_tu_pipe3d', 'xinv_tu_pipe3d.hip', ['-mllvm', '-amdgpu-sched-strategy=iterative-minreg']),
    # the biharmonic update is 51 dependent-chain flops per point and colour stage at one or two wavefronts per SIMD: the
    # default (occupancy-first) scheduler serialises the chains; max-ILP interleaves them (Munk 2000x2000: 1.04 -> 1.15e11)
    ('xinv_tu_bih', 'xinv_tu_bih.hip', ['-mllvm', '-amdgpu-sched-strategy=max-ilp']),
]
# XINV_VARIANT_UNITS="xinv_tu_fused3d,..." (with XINV_BUILD_TAG): only these units are compiled with the extra flags; every
# other object is taken from the shipped build's build/obj (a variant of one kernel family links in seconds)
VARIANT_UNITS = [u for u in os.environ.get('XINV_VARIANT_UNITS', '').split(',') if u] if TAG else []
MAIN_OBJ = os.path.join(HERE, '..', 'build', 'obj')
SOURCES = sorted({u[1] for u in UNITS})
# -ffp-contract=off: no FMA contraction, so device results are bitwise those of the
# CPU restatement of the same sweep ordering (see DESIGN.md "Arithmetic").
# -amdgpu-scalarize-global-loads=0: the compiler must not turn a uniform load from GLOBAL memory into an
# s_load.  The scalar data cache is not coherent with vector stores and was seen serving a previous solve's
# data from a reused workspace address; with this switch only loads the source spells out
# through the constant address space (kernel arguments; k_pipe2d's per-row records, behind its
# s_dcache_inv) use the scalar unit.  tools/smem_audit.py checks the built objects for exactly that.
BASE_FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC',
              '-mllvm', '-amdgpu-scalarize-global-loads=0']


def hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found')


def _headers():
    hs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith('.h')]
    hs.append(os.path.join(HERE, '..', 'include', 'xinv.h'))
    return hs


def _stamp(src, flags):
    """Content hash of everything a unit depends on (its source, every header, its flags)."""
    h = hashlib.sha256()
    for f in [src] + _headers():
        with open(f, 'rb') as fh:
            h.update(fh.read())
    h.update(' '.join(flags).encode())
    return h.hexdigest()


def source_hash():
    """Content hash (16 hex digits) of everything the library is built from: every file of csrc/ and include/xinv.h.
    profiles/traffic.json entries carry the hash of the tree they were profiled on; bench.py reports a counter figure
    only while it still matches (a kernel change that forgets to re-profile must not carry stale bytes forward)."""
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(('.h', '.hip'))]
    files.append(os.path.join(HERE, '..', 'include', 'xinv.h'))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _units():
    return [u for u in UNITS if os.path.exists(os.path.join(CSRC, u[1]))]


def _paths(tag):
    so = os.path.join(HERE, 'libxinv_hip.so') if not tag else os.path.join(HERE, '..', 'build', 'libxinv_%s.so' % tag)
    obj = os.path.join(HERE, '..', 'build', 'obj' + ('_' + tag if tag else ''))
    return so, obj


def stale(tag=TAG, extra=EXTRA, variant_units=VARIANT_UNITS):
    so, obj = _paths(tag)
    if not os.path.exists(so) or not os.path.exists(so + '.linkstamp'):
        return True
    if tag and os.path.getmtime(so) < os.path.getmtime(_paths('')[0]):
        return True                                     # (a variant links the shipped build's other objects)
    for name, src, uflags in _units():
        if variant_units and name not in variant_units:
            continue
        st = os.path.join(obj, name + '.stamp')
        if not os.path.exists(os.path.join(obj, name + '.o')) or not os.path.exists(st):
            return True
        if open(st).read() != _stamp(os.path.join(CSRC, src), BASE_FLAGS + list(extra) + uflags):
            return True
    return False


def build(force=False, verbose=False, jobs=None, tag=TAG, extra=EXTRA, variant_units=VARIANT_UNITS):
    """The shipped library (tag ''), or a variant build/libxinv_<tag>.so: `extra` flags on `variant_units` (every unit
    when empty), the other objects taken from the shipped build's build/obj."""
    so, obj = _paths(tag)
    variant_units = list(variant_units) if tag else []
    if not force and not stale(tag, extra, variant_units):
        return so
    os.makedirs(obj, exist_ok=True)
    cc = hipcc()
    todo = []
    for name, src, uflags in _units():
        if variant_units and name not in variant_units:
            continue
        srcp = os.path.join(CSRC, src)
        o, st = os.path.join(obj, name + '.o'), os.path.join(obj, name + '.stamp')
        flags = BASE_FLAGS + list(extra) + uflags
        stamp = _stamp(srcp, flags)
        if force or not os.path.exists(o) or not os.path.exists(st) or open(st).read() != stamp:
            todo.append((name, [cc] + flags + ['-c', srcp, '-o', o], st, stamp))

    def run(job):
        name, cmd, st, stamp = job
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(st, 'w') as fh:
            fh.write(stamp)

    jobs = jobs or min(len(todo) or 1, max(1, (os.cpu_count() or 2) - 1))
    with ThreadPoolExecutor(jobs) as ex:
        list(ex.map(run, todo))
    link = [cc, '--offload-arch=gfx950', '-shared', '-fPIC'] + \
           [os.path.join(obj if (not variant_units or u[0] in variant_units) else MAIN_OBJ, u[0] + '.o') for u in _units()] + \
           ['-o', so]
    if verbose:
        print(' '.join(link), flush=True)
    subprocess.check_call(link)
    with open(so + '.linkstamp', 'w') as fh:             # what was linked: fresh() compares it with the objects' stamps
        fh.write(_link_stamp(obj, variant_units))
    return so


def _link_stamp(obj, variant_units):
    """Hash of the stamps of every object a library is linked from (its own and, for a variant, the shipped build's)."""
    h = hashlib.sha256()
    for u in _units():
        d = obj if (not variant_units or u[0] in variant_units) else MAIN_OBJ
        st = os.path.join(d, u[0] + '.stamp')
        h.update(open(st).read().encode() if os.path.exists(st) else b'missing')
    return h.hexdigest()


def fresh(tag=TAG, extra=EXTRA, variant_units=VARIANT_UNITS):
    """True when the library `tag` was linked from objects compiled from the CURRENT sources (content hashes only: no
    file times, so the answer survives a copy of the tree to another box).  tests/test_gpu_watchdog.py asserts it for the
    test-hooks variant: a stale hooks library would exercise old kernels and old host code."""
    so, obj = _paths(tag)
    variant_units = list(variant_units) if tag else []
    if not os.path.exists(so) or not os.path.exists(so + '.linkstamp'):
        return False
    for name, src, uflags in _units():
        mine = not variant_units or name in variant_units
        d = obj if mine else MAIN_OBJ
        st = os.path.join(d, name + '.stamp')
        flags = BASE_FLAGS + (list(extra) if mine else []) + uflags
        if not os.path.exists(st) or open(st).read() != _stamp(os.path.join(CSRC, src), flags):
            return False
    return open(so + '.linkstamp').read() == _link_stamp(obj, variant_units)


# The test-hooks variant (tests/hooks_suite, run by tests/test_gpu_watchdog.py in a subprocess with XINV_SO set): the
# shipped library plus -DXINV_TEST_HOOKS=1 on the host driver and the kernel families the hooks live in (k_fused2d standard / general, k_pipe2d standard) -- a test
# can make one tile withhold its norm partial (a REAL reducer timeout, with a 30 ms watchdog) or leave a member in the
# state a timed-out reducer leaves behind.  The shipped library has neither hook nor environment switch.
HOOKS_TAG = 'hooks'
HOOKS_UNITS = ['xinv_hip', 'xinv_tu_fused2d_std', 'xinv_tu_fused2d_gen', 'xinv_tu_fused2d_genq', 'xinv_tu_pipe2d_std']
HOOKS_SO = os.path.join(HERE, '..', 'build', 'libxinv_%s.so' % HOOKS_TAG)


def build_hooks(force=False, verbose=False):
    build(force=False, verbose=verbose, tag='', extra=[], variant_units=[])        # (the objects it links)
    return build(force=force, verbose=verbose, tag=HOOKS_TAG, extra=['-DXINV_TEST_HOOKS=1'], variant_units=HOOKS_UNITS)


if __name__ == '__main__':
    j = [int(a[2:]) for a in sys.argv if a.startswith('-j') and a[2:].isdigit()]
    print(build(force='--force' in sys.argv, verbose=True, jobs=j[0] if j else None))
    if '--hooks' in sys.argv:
        print(build_hooks(force='--force' in sys.argv, verbose=True))
