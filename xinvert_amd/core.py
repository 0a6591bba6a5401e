"""Core inversion wrappers: inv_standard2D / inv_general2D / inv_standard3D.

Host-side mirror of the reference's xinvert/core.py:20-155, 374-444 -- same names, argument
order and error behaviour -- with one structural change: the reference's Python loop over the
non-core (time / level / member) axis (`for selDict in loop_noncore(F, dims)`, core.py:59,
129, 418) becomes ONE batched call into the HIP library, which keeps coefficient arrays that
do not vary along the batch axis resident once in HBM (batch stride 0) and solves every slice
concurrently.  There is no CPU fallback: a missing library or GPU raises.

Arrays may be `Field`s or ndarrays.  Coefficients are either core-shaped ([yc, xc] or
[zc, yc, xc]: shared by all slices) or shaped like F (one per slice).
"""
import itertools
import os

import numpy as np

from . import _lib
from .field import Field, LazyForcing, aligned, undef_as

# default undefined value (reference core.py:15)
_undeftmp = -9.99e8


def inv_standard2D(A, B, C, F, S, dims, iParams):
    """d/dy(A dS/dy + B dS/dx) + d/dx(B dS/dy + C dS/dx) = F   (reference core.py:88-155)."""
    if len(dims) != 2:
        raise Exception('2 dimensions are needed for inversion')
    return _solve('std2d', (A, B, C), F, S, dims, iParams)


def inv_standard2D_test(A, B, C, D, E, F, S, dims, iParams):
    """d/dy(A dS/dy + B dS/dx) + d/dx(C dS/dy + D dS/dx) + E S = F   (reference core.py:159-231;
    Fofonoff, Bretherton-Haidvogel)."""
    if len(dims) != 2:
        raise Exception('2 dimensions are needed for inversion')
    return _solve('std2dt', (A, B, C, D, E), F, S, dims, iParams)


def inv_general2D(A, B, C, D, E, F, G, S, dims, iParams):
    """A Syy + B Syx + C Sxx + D Sy + E Sx + F S = G   (reference core.py:374-444)."""
    if len(dims) != 2:
        raise Exception('2 dimensions are needed for inversion')
    return _solve('gen2d', (A, B, C, D, E, F), G, S, dims, iParams)


def inv_general2D_bih(A, B, C, D, E, F, G, H, I, J, S, dims, iParams):
    """A Syyyy + B Syyxx + C Sxxxx + D Syy + E Syx + F Sxx + G Sy + H Sx + I S = J
    (reference core.py:447-532; Munk / Stommel-Munk)."""
    if len(dims) != 2:
        raise Exception('2 dimensions are needed for inversion')
    return _solve('bih2d', (A, B, C, D, E, F, G, H, I), J, S, dims, iParams)


def inv_standard3D(A, B, C, F, S, dims, iParams):
    """d/dz(A dS/dz) + d/dy(B dS/dy) + d/dx(C dS/dx) = F   (reference core.py:20-85)."""
    if len(dims) != 3:
        raise Exception('3 dimensions are needed for inversion')
    return _solve('std3d', (A, B, C), F, S, dims, iParams)


def inv_general3D(A, B, C, D, E, F, G, H, S, dims, iParams):
    """A d2S/dz2 + B d2S/dy2 + C d2S/dx2 + D dS/dz + E dS/dy + F dS/dx + G S = H
    (reference core.py:294-371)."""
    if len(dims) != 3:
        raise Exception('3 dimensions are needed for inversion')
    return _solve('gen3d', (A, B, C, D, E, F, G), H, S, dims, iParams)


# ------------------------------------------------------------------------------ internals
def _vals(a):
    return a.values if isinstance(a, Field) or hasattr(a, 'values') else np.asarray(a)


def _batch_layout(F, dims):
    """Permutation putting the non-core axes first (in F's order) and the core axes last
    (in the order given by `dims`), plus the batch selectors for the info lines."""
    core_ax = [F.axis(d) for d in dims]
    batch_ax = [ax for ax in range(len(F.dims)) if ax not in core_ax]
    perm = batch_ax + core_ax
    bdims = [F.dims[ax] for ax in batch_ax]
    bshape = [F.shape[ax] for ax in batch_ax]
    return perm, bdims, bshape


def _prep_coef(c, F, perm, core_shape, nbatch, allow_null=False):
    """-> (contiguous float64 array, batch stride in elements, rowconst).

    (None, 0, False) for an identically-zero stride-0 view where the C-ABI accepts NULL (the
    cross coefficient B).  rowconst: the coefficient does not vary along x (a stride-0 view along
    the last core axis, as the lat-lon builders of apps.py produce): only its first column
    travels, [rows] per member, and the library expands it on the device."""
    labelled = hasattr(c, 'dims') and hasattr(c, 'values')
    if labelled:
        # labelled coefficient (Field / DataArray): line it up with F by dim NAME, as the
        # reference's xarray arithmetic does -- never by shape coincidence (square cores)
        v = np.broadcast_to(aligned(c, F), F.shape)
        if v.dtype not in (np.float64, np.float32):       # (float32 travels as float32, xinv_options.f32_mask; anything else is promoted here)
            v = v.astype(np.float64)
    else:
        v = np.asarray(_vals(c), dtype=np.float64)
    n = int(np.prod(core_shape))
    rows = n // core_shape[-1]
    if allow_null and v.size > 1 and all(st == 0 for st in v.strides) and v.flat[0] == 0.0:
        return None, 0, False
    if not labelled and v.shape == tuple(core_shape):     # bare ndarray in core layout [dims[0], dims[1], ...]
        if v.strides[-1] == 0 and core_shape[-1] > 1:
            return np.ascontiguousarray(v[..., 0]), 0, True
        return np.ascontiguousarray(v), 0, False
    if v.shape == F.shape:
        # a broadcast view over the batch axes (zero strides) is one shared slice
        t = np.transpose(v, perm)
        nb_axes = len(perm) - len(core_shape)
        rc = t.strides[-1] == 0 and core_shape[-1] > 1
        if all(t.strides[ax] == 0 or t.shape[ax] == 1 for ax in range(nb_axes)):
            idx = (0,) * nb_axes
            if rc:
                return np.ascontiguousarray(t[idx][..., 0]), 0, True
            return np.ascontiguousarray(t[idx]), 0, False
        if rc:
            return np.ascontiguousarray(t[..., 0]).reshape((nbatch, rows)), rows, True
        return np.ascontiguousarray(t).reshape((nbatch,) + tuple(core_shape)), n, False
    try:
        t = np.broadcast_to(v, F.shape)
    except ValueError:
        raise Exception('coefficient shape %r matches neither the core shape %r nor F %r'
                        % (v.shape, tuple(core_shape), F.shape))
    return _prep_coef(t, F, perm, core_shape, nbatch, allow_null)


_KIND_OF = {'inv_standard2D': 'std2d', 'inv_standard2D_test': 'std2dt', 'inv_general2D': 'gen2d',
            'inv_general2D_bih': 'bih2d', 'inv_standard3D': 'std3d', 'inv_general3D': 'gen3d'}


class Resident:
    """The batch of one `inv_*` call kept in HBM across solves: coefficient stack, forcing and S are
    uploaded ONCE, every `solve()` continues from the current S (the kernels are in place and
    restartable) and only the flags -- and, on `values()`, S -- cross PCIe.  For callers that solve
    the same operator repeatedly, as `apps.animate_iteration` does frame after frame (reference
    apps.py:1031-1044: one `invt_func(*coeffs, maskF, initS, dims, iParams)` per frame).  The solves run on a resident
    plan (include/xinv.h, xinv_plan_*): detection passes, per-row records and tile lists are built on the first solve
    and reused by every later one; the coefficient stack must not change while the object lives."""

    def __init__(self, inv_name, coefs, F, S, dims, iParams):
        from .resident import ResidentProblem
        kind = _KIND_OF[inv_name]
        if not isinstance(F, Field) or not isinstance(S, Field):
            raise Exception('forcing and solution must be Field objects (see xinvert_amd.field)')
        self.F, self.dims = F, dims
        self.perm, _, self.bshape = _batch_layout(F, dims)
        core_shape = tuple(F.shape[F.axis(d)] for d in dims)
        self.core_shape = core_shape
        nbatch = int(np.prod(self.bshape)) if self.bshape else 1
        tr = lambda v: np.ascontiguousarray(np.transpose(np.asarray(v, dtype=np.float64), self.perm)
                                            ).reshape((nbatch,) + core_shape)
        cs, shared = [], []
        for k, c in enumerate(coefs):
            a, st, rc = _prep_coef(c, F, self.perm, core_shape, nbatch, allow_null=(k == 1 and kind in ('std2d', 'gen2d')))
            if a is None:
                a = np.zeros(core_shape)                              # ResidentProblem passes it as NULL again
            elif rc:                                                   # one value per row STAYS one value per row: a stride-0 view
                a = a.reshape(((nbatch,) if st else ()) + core_shape[:-1])    # along x, which ResidentProblem uploads as rows
                a = np.broadcast_to(a[..., None], a.shape + (core_shape[-1],))   # (the plan expands them in HBM: rowconst_mask)
            if st == 0:
                shared.append(k)
            cs.append(a)
        cs.append(tr(F.values))
        ip = iParams
        p = dict(kind=kind, S0=tr(S.values), coefs=cs, shared=tuple(shared), undef=_undeftmp,
                 optArg=float(ip['optArg']), delxSqr=float(ip['del1Sqr']))
        BCs = list(ip['BCs'])
        if len(dims) == 2:
            p.update(yc=int(ip['gc2']), xc=int(ip['gc1']), dely=float(ip['del2']), delx=float(ip['del1']),
                     BCy=BCs[0], BCx=BCs[1], ratio=float(ip['ratio']), ratioQtr=float(ip['ratioQtr']),
                     ratioSqr=float(ip['ratioSqr']))
            if kind == 'bih2d':
                p.update(delxSSr=float(ip['del1SSr']), delxTr=float(ip['del1Tr']), ratioSSr=float(ip['ratioSSr']))
        else:
            p.update(zc=int(ip['gc3']), yc=int(ip['gc2']), xc=int(ip['gc1']), delz=float(ip['del3']),
                     dely=float(ip['del2']), delx=float(ip['del1']), BCz=BCs[0], BCy=BCs[1], BCx=BCs[2],
                     ratio2Sqr=float(ip['ratio2Sqr']), ratio1Sqr=float(ip['ratio1Sqr']))
            if kind == 'gen3d':
                p.update(ratio2=float(ip['ratio2']), ratio1=float(ip['ratio1']))
        self.iParams = iParams
        dev = int(iParams.get('device', -1))
        if dev < 0:
            import torch
            dev = torch.cuda.current_device()
        # (iParams['resident_plan'] = False: every solve re-derives what a plan keeps -- rounds 1-4, for comparisons)
        self.rp = ResidentProblem(p, device=dev, plan=bool(iParams.get('resident_plan', True)))

    def solve(self, mxLoop, tolerance, **opt):
        """One more call of the hot path on the resident batch -> flags [nbatch, 3].  Engine options the
        caller put into iParams (engine_path, sweeps_per_launch, check_every) apply as in the inv_* calls;
        flags and statistics are left in iParams as those calls leave them."""
        o = dict(path=int(self.iParams.get('engine_path', 0)),
                 sweeps_per_launch=int(self.iParams.get('sweeps_per_launch', 0)),
                 check_every=int(self.iParams.get('check_every', 0)))
        o.update(opt)
        fl, self.stats = self.rp.solve(mxLoop, tolerance, **o)
        self.iParams['flags'] = np.array(fl if self.rp.nb > 1 else fl[0], copy=True)
        self.iParams['stats'] = self.stats
        return fl

    def solve_frames(self, n, mxLoop, tolerance, **opt):
        """`n` restarts of the solve, every one continuing from the last, their S snapshots kept in HBM (keep_frames first):
        queued behind each other without a host round trip (ResidentProblem.solve_frames).  -> flags [n, nbatch, 3]; the
        snapshots join those of snapshot() in frames()."""
        o = dict(path=int(self.iParams.get('engine_path', 0)),
                 sweeps_per_launch=int(self.iParams.get('sweeps_per_launch', 0)),
                 check_every=int(self.iParams.get('check_every', 0)))
        o.update(opt)
        if not self.rp.use_plan:                             # (iParams['resident_plan'] = False: frame by frame, as rounds 1-4)
            out = []
            for _ in range(int(n)):
                out.append(np.array(self.solve(mxLoop, tolerance, **opt), copy=True))
                self.snapshot()
            return np.stack(out)
        out = []
        left = int(n)
        while left > 0:
            if self._nframe == self._frames.shape[0]:
                self._flush_frames()
            k = min(left, int(self._frames.shape[0]) - self._nframe)
            fl, self.stats = self.rp.solve_frames(self._frames[self._nframe:self._nframe + k], mxLoop, tolerance, **o)
            self._nframe += k
            left -= k
            out.append(fl)
        fl = np.concatenate(out)
        self.iParams['flags'] = np.array(fl[-1] if self.rp.nb > 1 else fl[-1][0], copy=True)
        self.iParams['stats'] = self.stats
        return fl

    def values(self):
        """S in the forcing's axis order (a fresh host array)."""
        out = self.rp.result().reshape(tuple(self.bshape) + self.core_shape)
        return np.ascontiguousarray(np.transpose(out, np.argsort(self.perm)))

    def keep_frames(self, n, max_bytes=None):
        """Room in HBM for up to `n` snapshots of S (apps.animate_iteration: the frames stay on the device -- a copy queued
        behind each solve -- and cross PCIe in blocks instead of one blocking download per frame).  The block is capped at
        `max_bytes` (default: a quarter of the HBM that is free right now, so that the solver's rotation twins of S still
        fit) and at what the allocator grants; a full block is flushed to the host and reused."""
        import torch
        per = int(self.rp.S.numel()) * self.rp.S.element_size()
        if max_bytes is None:
            try:
                max_bytes = torch.cuda.mem_get_info(self.rp.S.device)[0] // 4
            except Exception:
                max_bytes = 1 << 30
        cap = int(max(1, min(int(n), max_bytes // max(per, 1))))
        self._frames = None
        while self._frames is None:
            try:
                self._frames = torch.empty((cap,) + tuple(self.rp.S.shape), dtype=self.rp.S.dtype, device=self.rp.S.device)
            except RuntimeError:                             # (out of memory: a smaller block, down to one frame at a time)
                if cap == 1:
                    raise
                cap = max(1, cap // 2)
        self._nframe = 0
        self._flushed = []

    def _flush_frames(self):
        if self._nframe:
            self._flushed.append(self._frames[:self._nframe].cpu().numpy())
            self._nframe = 0

    def snapshot(self):
        self.rp._join()                                      # (a solve on a non-current stream: the copy waits for it)
        if self._nframe == self._frames.shape[0]:
            self._flush_frames()
        self._frames[self._nframe].copy_(self.rp.S)      # (device to device, on the stream the solve ran on)
        self._nframe += 1

    def frames(self):
        """The snapshots taken so far, [n, ...] in the forcing's axis order."""
        self._flush_frames()
        blocks, self._flushed = self._flushed, []
        out = np.concatenate(blocks) if blocks else np.empty((0,) + tuple(self.rp.S.shape))
        self._flushed = [out] if len(out) else []
        out = out.reshape((out.shape[0],) + tuple(self.bshape) + self.core_shape)
        inv = np.argsort(self.perm)
        return np.ascontiguousarray(np.transpose(out, (0,) + tuple(1 + a for a in inv)))


# in-process multi-GPU is the default only when every device gets a worthwhile share: below this much
# per-member data per device, the contexts, workspaces and re-uploaded coefficient stacks cost more than
# the extra GPUs give (a 3-slice 73 x 144 call stays on the caller's device)
_MULTI_GPU_MIN_BYTES_PER_DEVICE = 256 << 20


def _device_list(iParams, nbatch, batch_bytes=0):
    """GPUs the batch axis is split over inside this one call (contiguous blocks, the order of the
    reference's slice loop core.py:129).  iParams['devices']: a list of ordinals or 'all' -- the explicit
    form.  Unset: the caller's current device, unless the batch is large enough to give every visible GPU
    at least _MULTI_GPU_MIN_BYTES_PER_DEVICE of per-member data; never when one device was asked for
    (iParams['device']), when the process is one rank of a one-process-per-GPU job (WORLD_SIZE > 1:
    xinvert_amd.dist shards instead), or for a single slice."""
    d = iParams.get('devices')
    if d is not None:
        return d
    if nbatch <= 1 or iParams.get('device') is not None or int(os.environ.get('WORLD_SIZE', '1')) > 1:
        return None
    if batch_bytes < _MULTI_GPU_MIN_BYTES_PER_DEVICE * 2:     # (decided by the size alone: no device query)
        return None
    ndev = _lib.load().xinv_device_count()
    if ndev < 2:
        return None
    use = int(min(ndev, nbatch, batch_bytes // _MULTI_GPU_MIN_BYTES_PER_DEVICE))
    return list(range(use)) if use >= 2 else None


def _info(sel):
    """Selector text of the reference's per-slice line (core.py:141-145)."""
    return (str(sel).replace('numpy.datetime64(', '').replace('numpy.timedelta64(', '')
            .replace('np.float64(', '').replace('np.int64(', '').replace('np.float32(', '')
            .replace(')', '').replace('\'', '').replace('.000000000', ''))


_PINNED_RESULT_MAX_BYTES = 256 << 20


def _result_empty(shape, dtype, iParams):
    """The array a device-born solution lands in.  Out of torch's PINNED host pool when it is there (and the result is at
    most 256 MiB: pinned blocks stay in that pool): the library finds the array pinned (hipPointerGetAttributes) and the
    DMA engine writes it directly -- no staging copy through the library's ring on the way down (3600 x 1800: 0.7 ms of
    8.4).  iParams['pinned_result'] = False keeps plain pageable memory."""
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    if iParams.get('pinned_result', True) and (1 << 20) <= nbytes <= _PINNED_RESULT_MAX_BYTES:
        try:
            import torch
            t = torch.empty(tuple(shape), dtype=torch.float32 if np.dtype(dtype) == np.float32 else torch.float64,
                            pin_memory=True)
            return t.numpy()                                 # (the array keeps the tensor -- and its pinned block -- alive)
        except Exception:
            pass
    return np.empty(shape, dtype=dtype)


def _solve(kind, coefs, F, S, dims, iParams):
    if not isinstance(F, Field) or not (isinstance(S, Field) or S is None):
        raise Exception('forcing and solution must be Field objects (see xinvert_amd.field)')
    perm, bdims, bshape = _batch_layout(F, dims)
    core_shape = tuple(F.shape[F.axis(d)] for d in dims)
    nbatch = int(np.prod(bshape)) if bshape else 1
    n = int(np.prod(core_shape))
    if nbatch == 0:                # an empty non-core axis: the reference's loop_noncore yields nothing
        return S if S is not None else F.like(np.zeros(F.shape))
    L = _lib.require_gpu()

    # A lazy forcing travels as the caller's raw array; masking, the per-row scale, the zero initial
    # guess and the output de-mask then happen on the device (xinv_options.prep_flags).  Conditions:
    # the scale runs along the y core dim (row index), the forcing is the last array of the call.
    prep = None
    if isinstance(F, LazyForcing) and S is None and \
            (F.scale is None or (F.scale_dim in dims and list(dims).index(F.scale_dim) == len(dims) - 2)):
        # (the mask value as the raw array's own dtype stores it: a float32 forcing carries float32(undef), which the
        #  device pass compares after promotion -- ADVICE r4; the reference compares in float32, apps.py:2124-2128)
        prep = dict(mask='nan' if np.isnan(F.undef_in) else undef_as(np.asarray(F.raw).dtype, F.undef_in),
                    rowscale=F.scale, s_zero=True, demask=iParams['undef'])
        Fsrc = F.raw
        Sv = None                                            # (allocated below, once its dtype is known)
    else:
        s_created = S is None
        if S is None:
            S = F.like(np.zeros(F.shape))
        Fsrc = F.values
        Sv = np.ascontiguousarray(np.transpose(np.asarray(S.values, dtype=np.float64), perm)
                                  ).reshape((nbatch,) + core_shape)
    # A float32 forcing (every dataset the reference ships is float32, tests/test_Poisson.py:14-24) travels as float32 --
    # half the bytes over PCIe -- and is promoted on the device: the same float64 values a host-side promotion gives
    # (xinv_options.f32_mask).  iParams['float32_out']: the solution also comes back as float32, the dtype the
    # reference returns for float32 input (its initS is zeros_like(F)).
    f32_in = np.asarray(Fsrc).dtype == np.float32 and not iParams.get('no_float32_upload')
    # (float32 out: only where no float64 first guess would be rounded -- S created here, or handed in as float32)
    f32_out = bool(iParams.get('float32_out')) and (prep is not None or s_created or np.asarray(S.values).dtype == np.float32)
    Fv = np.ascontiguousarray(np.transpose(np.asarray(Fsrc, dtype=np.float32 if f32_in else np.float64), perm)
                              ).reshape((nbatch,) + core_shape)
    if prep is not None:
        Sv = _result_empty((nbatch,) + core_shape, np.float32 if f32_out else np.float64, iParams)
    elif f32_out:
        Sv = Sv.astype(np.float32)
    arrs, strides = [Sv], [n]
    rowconst = 0
    for k, c in enumerate(coefs):
        a, st, rc = _prep_coef(c, F, perm, core_shape, nbatch,
                               allow_null=(k == 1 and kind in ('std2d', 'gen2d')))
        arrs.append(a)
        strides.append(st)
        rowconst |= (1 << k) if rc else 0
    arrs.append(Fv)
    strides.append(n)

    flags = np.tile(np.array([0.0, 1.0, 0.0]), (nbatch, 1))
    BCs = [_lib.bc(b) for b in iParams['BCs']]
    # float32 arrays travel as float32 with their bit in xinv_options.f32_mask: derived from the dtypes of the arrays
    # actually handed over, and checked against what the code above meant to send (S, the forcing)
    f32_mask = sum(1 << k for k, a_ in enumerate(arrs) if a_ is not None and a_.dtype == np.float32)
    assert bool(f32_mask & 1) == bool(f32_out) and bool(f32_mask >> (len(coefs) + 1)) == bool(f32_in), f32_mask
    opt = _lib.options(device=int(iParams.get('device', -1)),
                       path=int(iParams.get('engine_path', 0)),
                       sweeps_per_launch=int(iParams.get('sweeps_per_launch', 0)),
                       check_every=int(iParams.get('check_every', 0)), rowconst_mask=rowconst,
                       host_chunk=int(iParams.get('host_chunk', 0)),
                       host_inflight=int(iParams.get('host_inflight', 0)),     # (chunk solves in flight; -1: the rolling batch of the 3-D form for any batch)
                       devices=_device_list(iParams, nbatch, sum(a.nbytes for a, st_ in zip(arrs, strides) if a is not None and st_)),
                       prep=prep,
                       f32_mask=f32_mask,                                        # bit 0: S; bit q + 1: coefficient q (the forcing last)
                       fma=1 if iParams.get('contracted') else 0)      # opt-in XINV_FLAG_FMA (include/xinv.h): NOT the reference's arithmetic
    st = _lib.strides_arg(strides)
    ptrs = [_lib.hptr(a, f32=bool((f32_mask >> k) & 1)) for k, a in enumerate(arrs)]
    mx, tol = int(iParams['mxLoop']), float(iParams['tolerance'])
    if kind == 'std2d':
        rc = L.xinv_standard_2d_f64_batched(
            *ptrs, nbatch, st, iParams['gc2'], iParams['gc1'],
            float(iParams['del2']), float(iParams['del1']), BCs[0], BCs[1],
            float(iParams['del1Sqr']), float(iParams['ratioQtr']), float(iParams['ratioSqr']),
            float(iParams['optArg']), _undeftmp, _lib.hptr(flags), mx, tol, opt)
    elif kind == 'gen2d':
        rc = L.xinv_general_2d_f64_batched(
            *ptrs, nbatch, st, iParams['gc2'], iParams['gc1'],
            float(iParams['del2']), float(iParams['del1']), BCs[0], BCs[1],
            float(iParams['del1Sqr']), float(iParams['ratio']), float(iParams['ratioQtr']),
            float(iParams['ratioSqr']), float(iParams['optArg']), _undeftmp,
            _lib.hptr(flags), mx, tol, opt)
    elif kind == 'std2dt':
        rc = L.xinv_standard_2d_test_f64_batched(
            *ptrs, nbatch, st, iParams['gc2'], iParams['gc1'],
            float(iParams['del2']), float(iParams['del1']), BCs[0], BCs[1],
            float(iParams['del1Sqr']), float(iParams['ratioQtr']), float(iParams['ratioSqr']),
            float(iParams['optArg']), _undeftmp, _lib.hptr(flags), mx, tol, opt)
    elif kind == 'bih2d':
        rc = L.xinv_general_bih_2d_f64_batched(
            *ptrs, nbatch, st, iParams['gc2'], iParams['gc1'],
            float(iParams['del2']), float(iParams['del1']), BCs[0], BCs[1],
            float(iParams['del1SSr']), float(iParams['del1Tr']), float(iParams['del1Sqr']),
            float(iParams['ratio']), float(iParams['ratioSSr']), float(iParams['ratioQtr']),
            float(iParams['ratioSqr']), float(iParams['optArg']), _undeftmp,
            _lib.hptr(flags), mx, tol, opt)
    elif kind == 'gen3d':
        rc = L.xinv_general_3d_f64_batched(
            *ptrs, nbatch, st, iParams['gc3'], iParams['gc2'], iParams['gc1'],
            float(iParams['del3']), float(iParams['del2']), float(iParams['del1']),
            BCs[0], BCs[1], BCs[2], float(iParams['del1Sqr']), float(iParams['ratio2']),
            float(iParams['ratio1']), float(iParams['ratio2Sqr']), float(iParams['ratio1Sqr']),
            float(iParams['optArg']), _undeftmp, _lib.hptr(flags), mx, tol, opt)
    else:
        rc = L.xinv_standard_3d_f64_batched(
            *ptrs, nbatch, st, iParams['gc3'], iParams['gc2'], iParams['gc1'],
            float(iParams['del3']), float(iParams['del2']), float(iParams['del1']),
            BCs[0], BCs[1], BCs[2], float(iParams['del1Sqr']),
            float(iParams['ratio2Sqr']), float(iParams['ratio1Sqr']),
            float(iParams['optArg']), _undeftmp, _lib.hptr(flags), mx, tol, opt)
    _lib.check(rc)

    # in place on S, as the reference (S.loc[sel].values are views of initS, apps.py:2159)
    inv = np.argsort(perm)
    out = np.transpose(Sv.reshape(tuple(bshape) + core_shape), inv)
    if prep is not None:                    # the solution was born on the device: no copy into an initS
        S = F.like(out if out.flags.c_contiguous else np.ascontiguousarray(out))
        iParams['_demasked'] = True
    elif S.values.dtype == out.dtype and S.values.flags.writeable:
        S.values[...] = out
    else:
        S.values = np.ascontiguousarray(out)

    iParams['flags'] = flags if nbatch > 1 else flags[0]
    iParams['stats'] = _lib.last_stats()
    if iParams.get('printInfo', True):
        coords = [np.asarray(F[d]) for d in bdims]
        sels = itertools.product(*coords) if bdims else [()]
        for m, idx in enumerate(sels):
            info = _info(dict(zip(bdims, idx)))
            tail = ' (overflows!)' if flags[m, 0] else ''
            print(info + ' loops {0:4.0f} and tolerance is {1:e}'.format(flags[m, 2], flags[m, 1])
                  + tail)
    return S
