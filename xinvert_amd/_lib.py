"""ctypes binding of libxinv_hip.so (C-ABI declared in include/xinv.h).

This is the ONLY compute path of the package: if the HIP library cannot be loaded, or it
reports no GPU, every solve raises -- there is no CPU fallback.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.environ.get('XINV_SO') or os.path.join(HERE, 'libxinv_hip.so')

BC_CODES = {'fixed': 0, 'extend': 1, 'periodic': 2}
PATH_AUTO, PATH_COLOUR, PATH_FUSED = 0, 1, 2
PREP_MASK_NAN, PREP_MASK_VALUE, PREP_ROWSCALE, PREP_S_ZERO, PREP_DEMASK = 1, 2, 4, 8, 16     # XINV_PREP_*
MAX_DEVICES = 16                      # XINV_MAX_DEVICES

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int64)
_i64, _f64, _int, _vp = ctypes.c_int64, ctypes.c_double, ctypes.c_int, ctypes.c_void_p


class XinvOptions(ctypes.Structure):
    _fields_ = [('device', ctypes.c_int32), ('path', ctypes.c_int32),
                ('sweeps_per_launch', ctypes.c_int32), ('check_every', ctypes.c_int32),
                ('rows_per_tile', ctypes.c_int32), ('timing', ctypes.c_int32),
                ('flags', ctypes.c_int32), ('rowconst_mask', ctypes.c_int32),
                ('host_chunk', ctypes.c_int32), ('ndev', ctypes.c_int32),
                ('device_ids', ctypes.c_int32 * MAX_DEVICES),
                ('prep_flags', ctypes.c_int32), ('f32_mask', ctypes.c_int32),
                ('prep_undef', ctypes.c_double), ('demask_value', ctypes.c_double),
                ('prep_rowscale', ctypes.POINTER(ctypes.c_double)),
                ('lanes', ctypes.c_int32), ('norm_lag', ctypes.c_int32),
                ('pipe_fr', ctypes.c_int32), ('graph', ctypes.c_int32),
                ('cu_count', ctypes.c_int32), ('host_inflight', ctypes.c_int32)]


class XinvStats(ctypes.Structure):
    _fields_ = [('path', ctypes.c_int32), ('colours', ctypes.c_int32),
                ('sweeps_per_launch', ctypes.c_int32), ('rows_per_tile', ctypes.c_int32),
                ('xuniform_mask', ctypes.c_int32), ('masked_tile_pct', ctypes.c_int32),
                ('sweep_launches', ctypes.c_int64), ('sweeps_max', ctypes.c_int64),
                ('sweep_ms', ctypes.c_double), ('h2d_ms', ctypes.c_double),
                ('d2h_ms', ctypes.c_double), ('wall_ms', ctypes.c_double),
                ('host_chunks', ctypes.c_int32), ('devices', ctypes.c_int32),
                ('pipelined', ctypes.c_int32), ('masked_tile_ppm', ctypes.c_int32),
                ('recovered_members', ctypes.c_int32), ('lanes', ctypes.c_int32),
                ('planned', ctypes.c_int32), ('point_factor', ctypes.c_int32), ('plan_ms', ctypes.c_double),
                ('k_chunks', ctypes.c_int32), ('cut_tiles', ctypes.c_int32),
                ('rolling', ctypes.c_int32), ('reserved_', ctypes.c_int32),
                ('launch_us_min', ctypes.c_double), ('launch_us_avg', ctypes.c_double),
                ('launch_us_max', ctypes.c_double)]


class XinvError(RuntimeError):
    pass


# every symbol include/xinv.h declares (tests check the library exports all of them)
EXPORTS = [
    'xinv_default_options', 'xinv_last_stats', 'xinv_last_error', 'xinv_device_count',
    'xinv_version', 'xinv_abi_sizes',
    'xinv_standard_2d_f64', 'xinv_general_2d_f64', 'xinv_standard_3d_f64',
    'xinv_standard_2d_f64_batched', 'xinv_general_2d_f64_batched',
    'xinv_standard_3d_f64_batched',
    'xinv_standard_2d_f64_dev', 'xinv_general_2d_f64_dev', 'xinv_standard_3d_f64_dev',
    'xinv_general_bih_2d_f64', 'xinv_general_bih_2d_f64_batched', 'xinv_general_bih_2d_f64_dev',
    'xinv_standard_2d_test_f64', 'xinv_standard_2d_test_f64_batched', 'xinv_standard_2d_test_f64_dev',
    'xinv_general_3d_f64', 'xinv_general_3d_f64_batched', 'xinv_general_3d_f64_dev',
    'xinv_gm_flow_f64_dev',
    'xinv_abs_norm_f64_dev',
    'xinv_plan_create_standard_2d_f64_dev', 'xinv_plan_create_general_2d_f64_dev',
    'xinv_plan_create_standard_3d_f64_dev', 'xinv_plan_create_general_3d_f64_dev',
    'xinv_plan_create_general_bih_2d_f64_dev', 'xinv_plan_create_standard_2d_test_f64_dev',
    'xinv_plan_solve_f64_dev', 'xinv_plan_solve_frames_f64_dev', 'xinv_plan_refresh', 'xinv_plan_destroy',
]

_lib = None


def load():
    """Load the HIP library (raises XinvError with a build hint when it is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO):
        raise XinvError('HIP extension %s is missing: run `python -m xinvert_amd.build` '
                        '(there is no CPU fallback)' % SO)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 / libhsa-runtime64.
    # If this library pulled in /opt/rocm's copy first, a later `import torch` would bring a
    # second runtime that finds no GPU.  Importing torch first (when present) makes the
    # dynamic linker resolve our dependency to the copy torch already mapped.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    try:
        L = ctypes.CDLL(SO)
    except OSError as e:
        raise XinvError('cannot load %s: %s' % (SO, e))
    _opt = ctypes.POINTER(XinvOptions)
    std2d_scal = [_i64, _i64, _f64, _f64, _int, _int, _f64, _f64, _f64, _f64, _f64, _dp, _i64, _f64]
    gen2d_scal = [_i64, _i64, _f64, _f64, _int, _int, _f64, _f64, _f64, _f64, _f64, _f64, _dp, _i64, _f64]
    std3d_scal = [_i64, _i64, _i64, _f64, _f64, _f64, _int, _int, _int, _f64, _f64, _f64, _f64,
                  _f64, _dp, _i64, _f64]
    L.xinv_standard_2d_f64.argtypes = [_dp] * 5 + std2d_scal
    L.xinv_general_2d_f64.argtypes = [_dp] * 8 + gen2d_scal
    L.xinv_standard_3d_f64.argtypes = [_dp] * 5 + std3d_scal
    L.xinv_standard_2d_f64_batched.argtypes = [_dp] * 5 + [_i64, _ip] + std2d_scal + [_opt]
    L.xinv_general_2d_f64_batched.argtypes = [_dp] * 8 + [_i64, _ip] + gen2d_scal + [_opt]
    L.xinv_standard_3d_f64_batched.argtypes = [_dp] * 5 + [_i64, _ip] + std3d_scal + [_opt]
    # *_dev: array arguments are device addresses (integers), flags is a host pointer
    std2d_dev = [_i64, _i64, _f64, _f64, _int, _int, _f64, _f64, _f64, _f64, _f64, _dp, _i64, _f64]
    L.xinv_standard_2d_f64_dev.argtypes = [_vp] * 5 + [_i64, _ip] + std2d_dev + [_opt, _vp]
    L.xinv_general_2d_f64_dev.argtypes = [_vp] * 8 + [_i64, _ip] + gen2d_scal + [_opt, _vp]
    L.xinv_standard_3d_f64_dev.argtypes = [_vp] * 5 + [_i64, _ip] + std3d_scal + [_opt, _vp]
    bih_scal = [_i64, _i64, _f64, _f64, _int, _int] + [_f64] * 9 + [_dp, _i64, _f64]
    L.xinv_general_bih_2d_f64.argtypes = [_dp] * 11 + bih_scal
    L.xinv_general_bih_2d_f64_batched.argtypes = [_dp] * 11 + [_i64, _ip] + bih_scal + [_opt]
    L.xinv_general_bih_2d_f64_dev.argtypes = [_vp] * 11 + [_i64, _ip] + bih_scal + [_opt, _vp]
    L.xinv_standard_2d_test_f64.argtypes = [_dp] * 7 + std2d_scal
    L.xinv_standard_2d_test_f64_batched.argtypes = [_dp] * 7 + [_i64, _ip] + std2d_scal + [_opt]
    L.xinv_standard_2d_test_f64_dev.argtypes = [_vp] * 7 + [_i64, _ip] + std2d_scal + [_opt, _vp]
    gen3d_scal = [_i64, _i64, _i64, _f64, _f64, _f64, _int, _int, _int] + [_f64] * 7 + [_dp, _i64, _f64]
    L.xinv_general_3d_f64.argtypes = [_dp] * 9 + gen3d_scal
    L.xinv_general_3d_f64_batched.argtypes = [_dp] * 9 + [_i64, _ip] + gen3d_scal + [_opt]
    L.xinv_general_3d_f64_dev.argtypes = [_vp] * 9 + [_i64, _ip] + gen3d_scal + [_opt, _vp]
    L.xinv_gm_flow_f64_dev.argtypes = [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _int, _int, _vp, _f64, _int, _vp]
    L.xinv_abs_norm_f64_dev.argtypes = [_vp, _i64, _f64, _dp, _vp]
    # resident plans: the *_dev argument lists without S / flags / mxLoop / tolerance, behind the handle's address
    _pp = ctypes.POINTER(_vp)
    no_tail = lambda scal: scal[:-3]                     # (drop flags, mxLoop, tolerance)
    L.xinv_plan_create_standard_2d_f64_dev.argtypes = [_pp] + [_vp] * 4 + [_i64, _ip] + no_tail(std2d_scal) + [_opt, _vp]
    L.xinv_plan_create_general_2d_f64_dev.argtypes = [_pp] + [_vp] * 7 + [_i64, _ip] + no_tail(gen2d_scal) + [_opt, _vp]
    L.xinv_plan_create_standard_3d_f64_dev.argtypes = [_pp] + [_vp] * 4 + [_i64, _ip] + no_tail(std3d_scal) + [_opt, _vp]
    L.xinv_plan_create_general_3d_f64_dev.argtypes = [_pp] + [_vp] * 8 + [_i64, _ip] + no_tail(gen3d_scal) + [_opt, _vp]
    L.xinv_plan_create_general_bih_2d_f64_dev.argtypes = [_pp] + [_vp] * 10 + [_i64, _ip] + no_tail(bih_scal) + [_opt, _vp]
    L.xinv_plan_create_standard_2d_test_f64_dev.argtypes = [_pp] + [_vp] * 6 + [_i64, _ip] + no_tail(std2d_scal) + [_opt, _vp]
    L.xinv_plan_solve_f64_dev.argtypes = [_vp, _vp, _dp, _i64, _f64, _vp]
    L.xinv_plan_solve_frames_f64_dev.argtypes = [_vp, _vp, _vp, _i64, _i64, _dp, _i64, _f64, _vp]
    L.xinv_plan_refresh.argtypes = [_vp, _vp]
    L.xinv_plan_destroy.argtypes = [_vp]
    for name in EXPORTS:
        getattr(L, name).restype = _int
    L.xinv_last_error.restype = ctypes.c_char_p
    L.xinv_default_options.restype = None
    L.xinv_default_options.argtypes = [_opt]
    L.xinv_last_stats.argtypes = [ctypes.POINTER(XinvStats)]
    # the structs above mirror include/xinv.h by hand: refuse a library whose layout differs
    so, ss = ctypes.c_int32(0), ctypes.c_int32(0)
    try:
        abi_sizes = L.xinv_abi_sizes
    except AttributeError:
        raise XinvError('%s predates xinv_abi_sizes (a stale prebuilt library): rebuild it with '
                        '`python -m xinvert_amd.build --force`' % SO) from None
    abi_sizes.restype = None
    abi_sizes(ctypes.byref(so), ctypes.byref(ss))
    if (so.value, ss.value) != (ctypes.sizeof(XinvOptions), ctypes.sizeof(XinvStats)):
        raise XinvError('%s was built from another include/xinv.h: xinv_options %d / xinv_stats %d bytes there, %d / %d '
                        'in xinvert_amd/_lib.py' % (SO, so.value, ss.value, ctypes.sizeof(XinvOptions),
                                                    ctypes.sizeof(XinvStats)))
    _lib = L
    return L


def require_gpu():
    L = load()
    if L.xinv_device_count() < 1:
        raise XinvError('libxinv_hip.so loaded but no MI355X/HIP device is visible '
                        '(there is no CPU fallback)')
    return L


def check(rc):
    if rc != 0:
        raise XinvError('xinv call failed (rc=%d): %s' % (rc, load().xinv_last_error().decode()))


def options(device=-1, path=PATH_AUTO, sweeps_per_launch=0, check_every=0, rows_per_tile=0,
            timing=0, no_xuniform=0, no_tile_skip=0, force_tile_skip=0, rowconst_mask=0,
            host_chunk=0, devices=None, prep=None, pin_host=0, no_pipe=0, fma=0, f32_mask=0,
            lanes=0, norm_lag=0, pipe_fr=0, graph=0, no_point_factor=0, cu_count=0, host_inflight=0):
    o = XinvOptions()
    load().xinv_default_options(ctypes.byref(o))
    o.device, o.path, o.sweeps_per_launch = device, path, sweeps_per_launch
    o.check_every, o.rows_per_tile, o.timing = check_every, rows_per_tile, timing
    # XINV_FLAG_NO_XUNIFORM | XINV_FLAG_NO_TILE_SKIP | XINV_FLAG_FORCE_TILE_SKIP
    o.flags = (1 if no_xuniform else 0) | (2 if no_tile_skip else 0) | (4 if force_tile_skip else 0) | \
              (8 if pin_host else 0) | (16 if no_pipe else 0) | (32 if fma else 0) | \
              (64 if no_point_factor else 0)      # ... | XINV_FLAG_PIN_HOST | XINV_FLAG_NO_PIPE | XINV_FLAG_FMA | XINV_FLAG_NO_POINT_FACTOR
    o.rowconst_mask = int(rowconst_mask)
    o.host_chunk = int(host_chunk)
    o.f32_mask = int(f32_mask)
    # expert overrides of the planner (0 = its own choice); the library reads no environment variable
    o.lanes, o.norm_lag, o.pipe_fr, o.graph = int(lanes), int(norm_lag), int(pipe_fr), int(graph)
    o.cu_count = int(cu_count)
    o.host_inflight = int(host_inflight)
    # prep: front-end passes on the device -- dict(mask='nan' | value, rowscale=vec | None,
    # s_zero=bool, demask=value | None); the row-scale array is kept alive on the options object
    if prep:
        fl = 0
        if isinstance(prep.get('mask'), str):
            fl |= PREP_MASK_NAN
        elif prep.get('mask') is not None:
            fl |= PREP_MASK_VALUE
            o.prep_undef = float(prep['mask'])
        if prep.get('rowscale') is not None:
            rs = np.ascontiguousarray(prep['rowscale'], dtype=np.float64)
            o._rowscale_keepalive = rs
            o.prep_rowscale = rs.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
            fl |= PREP_ROWSCALE
        if prep.get('s_zero'):
            fl |= PREP_S_ZERO
        if 'demask' in prep and prep['demask'] is not None:
            fl |= PREP_DEMASK
            o.demask_value = float(prep['demask'])
        o.prep_flags = fl
    # devices: None = the single device `device`; 'all' = every visible GPU; a list = those GPUs
    if devices is None:
        o.ndev = 0
    elif isinstance(devices, str):
        if devices != 'all':
            raise XinvError("devices must be None, 'all' or a list of device ordinals")
        o.ndev = -1
    else:
        devices = [int(d) for d in devices]
        if len(devices) > MAX_DEVICES:
            raise XinvError('at most %d devices' % MAX_DEVICES)
        o.ndev = len(devices)
        for i, d in enumerate(devices):
            o.device_ids[i] = d
    return o


def last_stats():
    s = XinvStats()
    load().xinv_last_stats(ctypes.byref(s))
    return {f: getattr(s, f) for f, _ in XinvStats._fields_}


def bc(b):
    if isinstance(b, str):
        if b not in BC_CODES:
            raise XinvError('unknown boundary condition %r' % (b,))
        return BC_CODES[b]
    return int(b)


def hptr(a, f32=False):
    """Host pointer of a C-contiguous float64 ndarray (None -> NULL).  `f32=True`: the array is float32 and the caller
    has set its bit in xinv_options.f32_mask (the C-ABI declares double* and reads floats there); any other dtype /
    flag combination raises -- a float32 array read as doubles is garbage and a read past its end."""
    if a is None:
        return None
    want = np.float32 if f32 else np.float64
    if a.dtype != want or not a.flags.c_contiguous:
        raise XinvError('need a C-contiguous %s array at the C-ABI here, got %s%s'
                        % (np.dtype(want).name, a.dtype.name, '' if a.flags.c_contiguous else ' (not contiguous)'))
    return ctypes.cast(a.ctypes.data, _dp)


def strides_arg(vals):
    return (ctypes.c_int64 * len(vals))(*[int(v) for v in vals])
