"""Stub-import of the reference kernels -- THIS CONTAINER ONLY (test infrastructure).

Loads /root/reference/xinvert/numbas.py by path with a 6-line stand-in for the
`numba` module whose `jit` decorator is the identity, so the reference kernels run
as plain Python with IEEE fp64 semantics (numba's nopython mode without fastmath
performs the same operations in the same order).  Nothing here is shipped to the
GPU box: the generator scripts that use it write numeric fixtures into
tests/golden/ and the reference tree never travels.

Only tests/golden/gen_golden.py and oracle self-checks run in the build container
may import this module.
"""
import importlib.util
import os
import sys
import types

REFERENCE_NUMBAS = '/root/reference/xinvert/numbas.py'


def available():
    return os.path.exists(REFERENCE_NUMBAS)


def load_reference_numbas():
    """Return the reference `numbas` module executed as pure Python."""
    if not available():
        raise RuntimeError('reference tree not present (expected only in the build container)')
    if 'numba' not in sys.modules:
        nb = types.ModuleType('numba')
        nb.jit = lambda *a, **k: (a[0] if (len(a) == 1 and callable(a[0]) and not k)
                                  else (lambda f: f))
        sys.modules['numba'] = nb
    spec = importlib.util.spec_from_file_location('ref_numbas', REFERENCE_NUMBAS)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
