"""ctypes front end of the C oracle (oracle/xinv_oracle.c).  TEST INFRASTRUCTURE ONLY.

Signatures mirror the reference kernels positionally (reference xinvert/numbas.py:215-219,
987-991, 15-19) with BC strings accepted as in the reference, plus a trailing `order`
(LEX = the reference's lexicographic sweep; COLOUR_* = the ordering the HIP kernels use).
"""
import ctypes
import os
import subprocess

import numpy as np

LEX, COLOUR_AUTO, COLOUR_2, COLOUR_4 = 0, 1, 2, 4
FMA = 0x100        # ORed into `order`: the contracted arithmetic of the HIP kernels' opt-in mode (xinv_oracle.c: XO_FMA)
BC_CODES = {'fixed': 0, 'extend': 1, 'periodic': 2}

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libxinv_oracle.so')
_lib = None

_dp = ctypes.POINTER(ctypes.c_double)
_i64, _f64, _int = ctypes.c_int64, ctypes.c_double, ctypes.c_int


def build(force=False):
    """Compile the C oracle with gcc (seconds)."""
    src = os.path.join(_HERE, 'xinv_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-s', '-C', _HERE, 'clean', 'all'])
    return _SO


_SO_NATIVE = os.path.join(_HERE, 'libxinv_oracle_native.so')
_portable = None


def use_native():
    """Switch to a -march=native build of the same source, compiled NOW on this machine (SURVEY.md
    8(d): the CPU baseline is the generous reading -- what numba/LLVM emits for the host CPU).
    Used by bench.py's cpu_baseline leg only; the travelling portable build stays the parity
    checker.  Returns False (and keeps the portable build) when gcc is not available."""
    global _lib, _portable
    src = os.path.join(_HERE, 'xinv_oracle.c')
    try:
        subprocess.check_call(['gcc', '-O3', '-march=native', '-std=c11', '-fPIC', '-ffp-contract=off',
                               '-fno-fast-math', '-shared', '-o', _SO_NATIVE, src, '-lm'])
    except Exception:
        return False
    if _portable is None and _lib is not None:
        _portable = _lib
    _lib = None
    _load(_SO_NATIVE)
    return True


def use_portable():
    global _lib, _portable
    if _portable is not None:
        _lib = _portable
    elif _lib is not None and getattr(_lib, '_xo_path', _SO) != _SO:
        _lib = None
    return lib()


def lib():
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _load(_SO)
    return _lib


def _load(path):
    global _lib, _portable
    if True:
        L = ctypes.CDLL(path)
        L._xo_path = path
        L.xo_standard_2d.restype = _int
        L.xo_standard_2d.argtypes = [_dp] * 5 + [_i64, _i64, _f64, _f64, _int, _int,
                                                 _f64, _f64, _f64, _f64, _f64, _dp, _i64,
                                                 _f64, _int]
        L.xo_general_2d.restype = _int
        L.xo_general_2d.argtypes = [_dp] * 8 + [_i64, _i64, _f64, _f64, _int, _int,
                                                _f64, _f64, _f64, _f64, _f64, _f64, _dp,
                                                _i64, _f64, _int]
        L.xo_standard_3d.restype = _int
        L.xo_standard_3d.argtypes = [_dp] * 5 + [_i64, _i64, _i64, _f64, _f64, _f64,
                                                 _int, _int, _int, _f64, _f64, _f64, _f64,
                                                 _f64, _dp, _i64, _f64, _int]
        L.xo_general_bih_2d.restype = _int
        L.xo_general_3d.restype = _int
        L.xo_general_3d.argtypes = [_dp] * 9 + [_i64, _i64, _i64, _f64, _f64, _f64, _int, _int, _int] + \
            [_f64] * 7 + [_dp, _i64, _f64, _int]
        L.xo_general_bih_2d.argtypes = [_dp] * 11 + [_i64, _i64, _f64, _f64, _int, _int] + \
            [_f64] * 9 + [_dp, _i64, _f64, _int]
        L.xo_standard_2d_test.restype = _int
        L.xo_standard_2d_test.argtypes = [_dp] * 7 + [_i64, _i64, _f64, _f64, _int, _int,
                                                      _f64, _f64, _f64, _f64, _f64, _dp, _i64,
                                                      _f64, _int]
        L.xo_abs_norm_2d.restype = _f64
        L.xo_abs_norm_2d.argtypes = [_dp, _i64, _i64, _f64]
        L.xo_abs_norm_3d.restype = _f64
        L.xo_abs_norm_3d.argtypes = [_dp, _i64, _i64, _i64, _f64]
        _lib = L
        if path == _SO:
            _portable = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp)


def _chk(a, shape, writable=False):
    if a.dtype != np.float64 or not a.flags.c_contiguous or a.shape != tuple(shape):
        raise ValueError('oracle needs C-contiguous float64 arrays of shape %r' % (shape,))
    if writable and not a.flags.writeable:
        raise ValueError('S must be writable')
    return a


def _bc(b):
    return BC_CODES[b] if isinstance(b, str) else int(b)


def standard_2d(S, A, B, C, F, yc, xc, dely, delx, BCy, BCx, delxSqr, ratioQtr, ratioSqr,
                optArg, undef, flags, mxLoop, tolerance, order=LEX):
    sh = (yc, xc)
    _chk(S, sh, True); [_chk(a, sh) for a in (A, B, C, F)]
    rc = lib().xo_standard_2d(_p(S), _p(A), _p(B), _p(C), _p(F), yc, xc, dely, delx,
                              _bc(BCy), _bc(BCx), delxSqr, ratioQtr, ratioSqr, optArg, undef,
                              _p(flags), mxLoop, tolerance, order)
    if rc:
        raise ValueError('oracle: bad arguments (rc=%d)' % rc)
    return S


def general_2d(S, A, B, C, D, E, F, G, yc, xc, dely, delx, BCy, BCx, delxSqr, ratio,
               ratioQtr, ratioSqr, optArg, undef, flags, mxLoop, tolerance, order=LEX):
    sh = (yc, xc)
    _chk(S, sh, True); [_chk(a, sh) for a in (A, B, C, D, E, F, G)]
    rc = lib().xo_general_2d(_p(S), _p(A), _p(B), _p(C), _p(D), _p(E), _p(F), _p(G), yc, xc,
                             dely, delx, _bc(BCy), _bc(BCx), delxSqr, ratio, ratioQtr,
                             ratioSqr, optArg, undef, _p(flags), mxLoop, tolerance, order)
    if rc:
        raise ValueError('oracle: bad arguments (rc=%d)' % rc)
    return S


def standard_3d(S, A, B, C, F, zc, yc, xc, delz, dely, delx, BCz, BCy, BCx, delxSqr,
                ratio2Sqr, ratio1Sqr, optArg, undef, flags, mxLoop, tolerance, order=LEX):
    sh = (zc, yc, xc)
    _chk(S, sh, True); [_chk(a, sh) for a in (A, B, C, F)]
    rc = lib().xo_standard_3d(_p(S), _p(A), _p(B), _p(C), _p(F), zc, yc, xc, delz, dely,
                              delx, _bc(BCz), _bc(BCy), _bc(BCx), delxSqr, ratio2Sqr,
                              ratio1Sqr, optArg, undef, _p(flags), mxLoop, tolerance, order)
    if rc:
        raise ValueError('oracle: bad arguments (rc=%d)' % rc)
    return S


def general_3d(S, A, B, C, D, E, F, G, H, zc, yc, xc, delz, dely, delx, BCz, BCy, BCx, delxSqr,
               ratio2, ratio1, ratio2Sqr, ratio1Sqr, optArg, undef, flags, mxLoop, tolerance,
               order=LEX):
    sh = (zc, yc, xc)
    _chk(S, sh, True); [_chk(a, sh) for a in (A, B, C, D, E, F, G, H)]
    rc = lib().xo_general_3d(_p(S), *[_p(a) for a in (A, B, C, D, E, F, G, H)], zc, yc, xc, delz,
                             dely, delx, _bc(BCz), _bc(BCy), _bc(BCx), delxSqr, ratio2, ratio1,
                             ratio2Sqr, ratio1Sqr, optArg, undef, _p(flags), mxLoop, tolerance,
                             order)
    if rc:
        raise ValueError('oracle: bad arguments (rc=%d)' % rc)
    return S


def general_bih_2d(S, A, B, C, D, E, F, G, H, I, J, yc, xc, dely, delx, BCy, BCx, delxSSr,
                   delxTr, delxSqr, ratio, ratioSSr, ratioQtr, ratioSqr, optArg, undef, flags,
                   mxLoop, tolerance, order=LEX):
    sh = (yc, xc)
    _chk(S, sh, True); [_chk(a, sh) for a in (A, B, C, D, E, F, G, H, I, J)]
    rc = lib().xo_general_bih_2d(_p(S), *[_p(a) for a in (A, B, C, D, E, F, G, H, I, J)], yc, xc,
                                 dely, delx, _bc(BCy), _bc(BCx), delxSSr, delxTr, delxSqr, ratio,
                                 ratioSSr, ratioQtr, ratioSqr, optArg, undef, _p(flags), mxLoop,
                                 tolerance, order)
    if rc:
        raise ValueError('oracle: bad arguments (rc=%d)' % rc)
    return S


def standard_2d_test(S, A, B, C, D, E, F, yc, xc, dely, delx, BCy, BCx, delxSqr, ratioQtr,
                     ratioSqr, optArg, undef, flags, mxLoop, tolerance, order=LEX):
    sh = (yc, xc)
    _chk(S, sh, True); [_chk(a, sh) for a in (A, B, C, D, E, F)]
    rc = lib().xo_standard_2d_test(_p(S), *[_p(a) for a in (A, B, C, D, E, F)], yc, xc, dely, delx,
                                   _bc(BCy), _bc(BCx), delxSqr, ratioQtr, ratioSqr, optArg, undef,
                                   _p(flags), mxLoop, tolerance, order)
    if rc:
        raise ValueError('oracle: bad arguments (rc=%d)' % rc)
    return S


def abs_norm(S, undef):
    S = np.ascontiguousarray(S, dtype=np.float64)
    if S.ndim == 2:
        return lib().xo_abs_norm_2d(_p(S), S.shape[0], S.shape[1], undef)
    return lib().xo_abs_norm_3d(_p(S), S.shape[0], S.shape[1], S.shape[2], undef)
