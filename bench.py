#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X SOR inversion engine.

Metric (BASELINE.json): SOR grid-points x iterations per second, fp64, on the 3600 x 1800 global
lat-lon Poisson problem with a land/sea mask (configs[1]).  One *step* is one complete hot-path
pass: `xinv_standard_2d_f64_dev` over one batch of synthetic input already resident in HBM,
running a fixed number of sweeps (tolerance = 0, mxLoop = sweeps - 1), norm + stopping rule
evaluated on the device after every sweep exactly as in production.  Every point of the grid is
counted in `value`, masked (land) points included; `value_active` and every roofline fraction count
only the tiles the kernel actually ran.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c4|c5] [--sweeps S] ...

--gpus N > 1 started WITHOUT a torchrun environment launches its own N ranks (torch.distributed.run,
rendezvous on 127.0.0.1) and fails loudly when fewer than N GPUs are visible; started under torchrun it
joins the job.  One rank per GPU, backend nccl (= RCCL); no data-path collective: the batch axis is split
in contiguous blocks and the per-slice flags are all-gathered after each step.  `n_gpus` in the JSON line
is the number of ranks that took part in that RCCL all-gather (`rccl_ranks`), not the number asked for.
--config c2 (default): every rank solves its own member(s) (weak scaling).
--config c4 / c5: the REAL batch of BASELINE configs[3] / [4] (64 Gill-Matsuno members / 120 omega
volumes) split over the ranks (strong scaling).
--inproc: additionally time the other multi-GPU mode: ONE process, host pointers,
xinv_options.ndev = N (one host thread per GPU inside the call); PCIe-inclusive, reported as `inproc`.

Extra objects in the ONE JSON line rank 0 prints (N = 1, c2):
  roofline       the resource that bounds the dominant kernel: the fp64 vector ALU.  achieved = point updates
                 of the tiles that RAN (skipped, fully masked tiles are not counted) x 16 fp64 operations /
                 mean launch duration (HIP events on the solve's stream over the timed region); peak = fp64
                 VALU issue rate WITHOUT FMA (the bit-exactness contract forbids contraction).
                 `alg_GBps` / `alg_frac` is SURVEY 8(d)'s 48 B/point figure over the same time -- a comparable
                 number, NOT an HBM fraction when > 1; `traffic` is the PMC-measured bytes per launch of the
                 same kernel variant (profiles/traffic.json; counters cannot be read in-process).
                 `hbm_frac` / `hbm_variant` repeat roofline_hbm's figures (the one true HBM fraction of the line) and
                 `configs` a compact copy of the per-configuration table.
  roofline_hbm   the HBM-bound variant north_star names on a working set far beyond the 256 MiB Infinity
                 Cache: 8 members with their own A, C, F (2.1 GB), one sweep per pass, every array streamed,
                 every tile run.  `achieved` prices the 40 B per point the variant really streams (S read +
                 write, A, C, F; B is identically zero and never read); `alg48_*` is SURVEY 8(d)'s 48 B figure.
  configs        one line per other BASELINE configuration (C1, C3 Stommel, C3 Munk, C4, C5) on this GPU at SURVEY
                 8(d)'s sweep counts (500; C5 200) and one GPU's share of the batch (C4 8 of 64, C5 15 of 120):
                 value, kernel, bound, fraction, alg_frac, PMC traffic, and a bitwise parity flag against the oracle.
  rank_values    (N > 1) every rank's own rate on its block; n1_value: rank 0 alone on its block with the others idle
                 (the N = 1 rate for the same per-GPU work); flags_sha256: digest of the gathered per-slice flags.
  parity         the timed solve repeated from the initial state and compared BIT FOR BIT with the oracle.
  cpu_baseline   the oracle's lexicographic sweep (the reference's execution model), -march=native build.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES = {'std2d': 48, 'gen2d': 72, 'std3d': 48, 'bih2d': 96}       # SURVEY.md section 8(d) / DESIGN 4.3
# fp64 operations of one point update as the kernels execute it (relaxation factor and F*delxSqr hoisted per
# row / per launch where the coefficients allow): std2d 4 sub + 4 mul + 2 sub + mul + add + sub + mul + add = 16;
# gen2d (hoisted) 25; std3d (hoisted) 21; bih2d (row scalars, B = E = 0) 45
UPD_FLOPS = {'std2d': 16, 'gen2d': 25, 'std3d': 21, 'bih2d': 45}
# first keys of `roofline`, in this order (the driver keeps 24)
ROOF_FIRST = ['bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_GBps', 'traffic_frac_of_hbm_peak',
              'hbm_frac', 'executed_over_useful', 'parity_bitwise', 'avg_launch_ms', 'launches', 'ms_outside_launches',
              'kernel', 'valu_frac', 'frac_of_fma_datasheet_peak',
              # (scalars of the other legs, so that they survive in the driver's record: VERDICT r5 items 4 and 5)
              'sustained_value', 'sustained_launch_drift', 'e2e_c2_invert_poisson_ms', 'e2e_c5x15_ms',
              'e2e_c5x15_vs_resident', 'e2e_c4x8_ms', 'src_sha',
              'configs', 'hbm_variant', 'alg_frac', 'alg_GBps', 'alg_bytes_per_launch',
              'streamed_bytes_per_point_sweep', 'active_tile_share', 'in_infinity_cache']
HBM_PEAK_GBS = 8000.0                                       # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
# fp64 vector ALU: 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz = 39.3 T operations/s (the 78.6 TFLOP/s
# datasheet figure counts an FMA as two; tools/fp64_peak.hip measures both on the box)
FP64_VALU_PEAK_TFLOPS = 39.3
FP64_VALU_MEASURED_TFLOPS = 34.0                             # profiles/r02_fp64_peak.txt (mul+add chains, no FMA)
FP64_FMA_SPEC_TFLOPS = 78.6


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=8)        # (untimed; 3 left the first box-fresh steps in the timed region: 6.6 / 6.9 / 7.0 against 7.0 / 6.9 / 7.1e11 with 20)
    ap.add_argument('--config', default='c2', choices=['c2', 'c4', 'c5'])
    ap.add_argument('--sweeps', type=int, default=0, help='SOR sweeps per step (0: SURVEY.md 8(d): 500 for c2/c4, 200 for c5)')
    ap.add_argument('--spl', type=int, default=0, help='sweeps fused per launch (0 = engine default)')
    ap.add_argument('--rows', type=int, default=0, help='rows per tile (0 = engine default)')
    ap.add_argument('--members', type=int, default=0, help='c2: batch members per GPU (default 1); c4/c5: total batch (default 64 / 120)')
    ap.add_argument('--ny', type=int, default=1800)
    ap.add_argument('--nx', type=int, default=3600)
    ap.add_argument('--grid', default='', help='c4: "ny,nx", c5: "nz,ny,nx" -- a reduced grid for the batched configurations '
                                               '(tests: the REAL 8-way shard shapes, 64 -> 8 members and 120 -> 15 volumes, at a size eight ranks on one GPU can hold)')
    ap.add_argument('--mask', default='continents', choices=['continents', 'coastline', 'none'],
                    help='c2 land mask: continent-size blobs (default, SURVEY 8(d)), coastline-scale, or none')
    ap.add_argument('--no-cpu', action='store_true', help='skip the cpu_baseline leg')
    ap.add_argument('--no-parity', action='store_true', help='skip the post-run oracle parity check')
    ap.add_argument('--no-hbm', action='store_true', help='skip the HBM-bound variant')
    ap.add_argument('--no-configs', action='store_true', help='skip the per-configuration lines')
    ap.add_argument('--inproc', action='store_true', help='also time the in-process multi-device mode (host pointers, ndev = N)')
    ap.add_argument('--no-e2e', action='store_true', help='skip the end-to-end (host arrays in, host arrays out) leg')
    ap.add_argument('--sustained-seconds', type=float, default=5.0)
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--parity-sweeps', type=int, default=0, help='sweeps of the parity check (0 = the timed count)')
    return ap.parse_args()


# ------------------------------------------------------------------ the oracle (checker / CPU baseline only)
def oracle_solve(q, mxLoop, tol, order):
    import oracle as orc
    S = np.array(q['S0'], dtype=np.float64, copy=True)
    fl = np.array([0., 1., 0.])
    c = [np.ascontiguousarray(a, dtype=np.float64) for a in q['coefs']]
    k = q['kind']
    if k == 'std2d':
        orc.standard_2d(S, *c, q['yc'], q['xc'], q['dely'], q['delx'], q['BCy'], q['BCx'], q['delxSqr'],
                        q['ratioQtr'], q['ratioSqr'], q['optArg'], q['undef'], fl, mxLoop, tol, order)
    elif k == 'gen2d':
        orc.general_2d(S, *c, q['yc'], q['xc'], q['dely'], q['delx'], q['BCy'], q['BCx'], q['delxSqr'],
                       q['ratio'], q['ratioQtr'], q['ratioSqr'], q['optArg'], q['undef'], fl, mxLoop, tol, order)
    elif k == 'bih2d':
        orc.general_bih_2d(S, *c, q['yc'], q['xc'], q['dely'], q['delx'], q['BCy'], q['BCx'], q['delxSSr'],
                           q['delxTr'], q['delxSqr'], q['ratio'], q['ratioSSr'], q['ratioQtr'], q['ratioSqr'],
                           q['optArg'], q['undef'], fl, mxLoop, tol, order)
    else:
        orc.standard_3d(S, *c, q['zc'], q['yc'], q['xc'], q['delz'], q['dely'], q['delx'], q['BCz'], q['BCy'],
                        q['BCx'], q['delxSqr'], q['ratio2Sqr'], q['ratio1Sqr'], q['optArg'], q['undef'], fl,
                        mxLoop, tol, order)
    return S, fl


def cpu_baseline(q, budget_s):
    """Time the oracle's lexicographic sweep (the reference's execution model: one slice, one
    core) on member 0 of the same workload for about `budget_s` seconds, with the -march=native
    build of the oracle made on THIS box; falls back to the travelling build."""
    import oracle as orc
    native = orc.use_native()
    npts = q['yc'] * q['xc']

    def run(nsweeps):
        t = time.perf_counter()
        oracle_solve(q, nsweeps - 1, 0.0, orc.LEX)
        return time.perf_counter() - t

    t2 = run(2)
    n = int(max(4, min(400, budget_s / max(t2 / 2, 1e-6))))
    t = run(n)
    flags = 'gcc -O3 -march=native -ffp-contract=off' if native else 'gcc -O3 -ffp-contract=off (no -march=native: gcc missing on the box)'
    out = {'value': npts * n / t, 'unit': 'point-sweeps/s', 'cores': 1, 'kind': 'port',
           'sample': '%d lexicographic sweeps of the full %dx%d slice (%.1f s), oracle/xinv_oracle.c %s'
                     % (n, q['yc'], q['xc'], t, flags)}
    # the most generous reading of the reference: every host core sweeping its
    # own slice of a batch at once (coefficients shared, one S per thread; ctypes drops the GIL)
    try:
        from concurrent.futures import ThreadPoolExecutor
        nthr = max(1, min(os.cpu_count() or 1, 256))
        if nthr > 1:
            with ThreadPoolExecutor(nthr) as ex:
                t0 = time.perf_counter()
                list(ex.map(lambda _: run(2), range(nthr)))          # probe: memory-bound, far from linear
                tp = time.perf_counter() - t0
                nsw = int(max(2, min(n, 2 * 10.0 / max(tp, 1e-6))))  # ~10 s of wall time
                t0 = time.perf_counter()
                list(ex.map(lambda _: run(nsw), range(nthr)))
                ta = time.perf_counter() - t0
            out['all_cores'] = {'value': npts * nsw * nthr / ta, 'unit': 'point-sweeps/s', 'cores': nthr,
                                'sample': '%d threads x %d sweeps, one slice each (%.1f s)' % (nthr, nsw, ta)}
    except Exception as e:                                   # the 1-core figure stands on its own
        out['all_cores'] = {'error': str(e)}
    orc.use_portable()
    return out


def oracle_parity(q, S_hip, flags_hip, sweeps, order=None):
    """Bitwise comparison of the HIP result with the oracle's coloured ordering (same input, same
    sweep count).  The checker, run after the timed region."""
    import oracle as orc
    orc.use_portable()                   # parity is defined against the -ffp-contract=off portable build
    order = orc.COLOUR_2 if order is None else order
    t = time.perf_counter()
    S, fl = oracle_solve(q, sweeps - 1, 0.0, order)
    dt = time.perf_counter() - t
    return {'bitwise': bool(np.array_equal(S, S_hip)), 'sweeps': int(sweeps),
            'mismatching_points': int((S != S_hip).sum()),
            'loop_equal': bool(fl[2] == flags_hip[2]),
            'flag1_abs_diff': float(abs(fl[1] - flags_hip[1])),
            'against': 'oracle/xinv_oracle.c coloured ordering, %.1f s on 1 core' % dt}


# ------------------------------------------------------------------ what a launch must do / move
def tile_model(s):
    """executed / useful point updates of the fused 2-D tiling actually used (recomputed halo rows
    and columns, pipeline steps rounded to the unroll)."""
    K, RY = s['sweeps_per_launch'], max(1, s['rows_per_tile'])
    if s['path'] != 2 or K < 1:
        return None
    if s.get('pipelined'):
        # k_pipe2d: wavefront p marches RY + 4K - 4p rows (p = 0..3) of 128 np columns, 128 np - 4K of them owned
        np_ = s['pipelined']
        return (RY + 4 * K - 6.0) * 128.0 * np_ / (RY * (128.0 * np_ - 4 * K))
    D = 2 * K + 2
    steps = -(-(RY + 4 * K) // D) * D
    UW = 128 - 4 * K
    return steps * 128.0 / (RY * UW)


def kernel_name(kind, s):
    K, um = s['sweeps_per_launch'], s['xuniform_mask']
    if s['path'] == 1:
        return 'colour-pass kernels (%d colours)' % s['colours']
    if kind == 'bih2d':
        vm = s.get('point_factor', 0)
        if vm:
            return 'k_fusedbih (one pass per sweep; %s as vector streams + the point-factor stream Q)' % \
                   ({1: 'A, C, D, F', 3: 'A (= C), D (= F)'}.get(vm, 'A..I'))
        return 'k_fusedbih (one pass per sweep, A..I and the relaxation factor as per-row records)'
    if kind == 'std3d':
        if K == 2:
            return 'k_pipe3d (two sweeps per pass, one per group of eight wavefronts; x-uniform mask=%d)' % um
        return 'k_fused3d<K=%d, x-uniform mask=%d>' % (K, um)
    model = {'std2d': 'Std2D', 'gen2d': 'Gen2D'}[kind]
    if s.get('pipelined'):
        return 'k_pipe2d<%s, NP=%d> (four sweeps per pass, one per wavefront; x-uniform mask=%d)' % (model, s['pipelined'], um)
    if s['colours'] == 4:
        return 'k_fused9<%s, K=%d>' % (model, K)
    pf = {0: '', 1: '; point-factor stream Q', 2: '; point-factor stream Q, C read out of A'}[s.get('point_factor', 0)]
    return 'k_fused2d<Fused%s, K=%d, x-uniform mask=%d%s>' % (model, K, um, pf)


def streamed_bytes_per_point_sweep(kind, s):
    """bytes the kernel variant must move per point-sweep: S read + S write + every coefficient array
    that is not a per-row scalar (and not an identically-zero B), 8 B each, over the K sweeps of a pass"""
    K, um = max(1, s['sweeps_per_launch']), s['xuniform_mask']
    if s['path'] != 2:
        return None
    if kind == 'std2d':
        nvec = 3 - bin(um & 7).count('1')                   # A, C, F
    elif kind == 'gen2d':
        nvec = 6 - bin(um & 63).count('1')                  # A, C, D, E, F, G
        nvec += {0: 0, 1: 1, 2: 0}[s.get('point_factor', 0)]   # (+ Q; - C when it is read out of A)
    elif kind == 'std3d':
        nvec = 4 - bin(um & 7).count('1')                   # A, B, C + forcing
    else:                                                   # biharmonic one-pass kernel: the forcing J (+ the vector streams + Q)
        nvec = 1 + {0: 0, 1: 5, 2: 10, 3: 3}[s.get('point_factor', 0)]
    return 8.0 * (2 + nvec) / K


def roofline_of(kind, s, pts_per_launch_all, avg_ms):
    """VALU and bandwidth fractions of the dominant kernel on the tiles that RAN; bound = the larger."""
    active = 1.0 - s.get('masked_tile_ppm', 0) / 1e6
    upd = pts_per_launch_all * active
    tf = UPD_FLOPS[kind] * upd / (avg_ms * 1e-3) / 1e12
    bpp = streamed_bytes_per_point_sweep(kind, s)
    gbs = (bpp * upd / (avg_ms * 1e-3) / 1e9) if bpp else None
    vf = tf / FP64_VALU_PEAK_TFLOPS
    hf = (gbs / HBM_PEAK_GBS) if gbs else 0.0
    r = {'valu_TFLOPs': tf, 'valu_frac': vf, 'streamed_GBps': gbs, 'streamed_frac_of_hbm_peak': hf if gbs else None,
         'streamed_bytes_per_point_sweep': bpp, 'active_tile_share': active, 'avg_launch_ms': avg_ms}
    # a launch whose buffers (S twice + the vector streams) fit the 256 MiB Infinity Cache is served by the fabric,
    # not by HBM (measured: 7 TB/s on the 259 MB K = 1 variant): its bytes are not priced against the HBM roof
    nvec = (bpp * max(1, s['sweeps_per_launch']) / 8.0 - 2) if bpp else 0
    ws = pts_per_launch_all / max(1.0, float(s['sweeps_per_launch'])) * 8.0 * (2 + nvec)
    r['working_set_bytes'] = ws
    r['in_infinity_cache'] = bool(ws < 230e6)
    if hf > vf and not r['in_infinity_cache']:
        r.update({'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': hf})
    else:
        r.update({'bound': 'valu_fp64', 'achieved': tf, 'peak': FP64_VALU_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': vf})
    return r


# ------------------------------------------------------------------ problems
def c2_problem(a, rank=0, members=1, shared=True):
    from xinvert_amd import synthetic
    mask = {'continents': True, 'coastline': 'coastline', 'none': False}[a.mask]
    p = synthetic.poisson_latlon(a.ny, a.nx, mask=mask, seed=synthetic.SEED + rank, members=members)
    if not shared:                                           # every member its own A and C (HBM leg: nothing shared
        p = dict(p)                                          # but the identically-zero B, which travels as NULL)
        p['coefs'] = [np.broadcast_to(c, (members,) + c.shape) if k in (0, 2) else c for k, c in enumerate(p['coefs'])]
        p['shared'] = (1,)
    return p


def build_problem(a, rank, world):
    """-> (problem dict restricted to this rank's members, total members over all ranks, scaling, name)."""
    from xinvert_amd import synthetic
    from xinvert_amd import dist as xdist
    if a.config == 'c2':
        nb = a.members or 1
        p = c2_problem(a, rank, nb)
        return p, nb * world, 'weak', 'invert_Poisson %dx%d lat-lon, land/sea mask (%s), periodic-x, fixed-y (BASELINE configs[1])' % (a.nx, a.ny, a.mask)
    total = a.members or (64 if a.config == 'c4' else 120)
    lo, hi = xdist.shard_range(total, rank, world)
    grid = [int(v) for v in a.grid.split(',')] if a.grid else None
    if a.config == 'c4':
        # every rank draws the same member list (same seed) and keeps its block
        ny, nx = grid if grid else (720, 1440)
        p = synthetic.gill_matsuno(ny, nx, total)
        p['S0'] = p['S0'][lo:hi]
        p['coefs'] = [c if k in p['shared'] else c[lo:hi] for k, c in enumerate(p['coefs'])]
        return p, total, 'strong', 'invert_GillMatsuno %dx%d, %d forcing members (BASELINE configs[3])' % (nx, ny, total)
    nz, ny, nx = grid if grid else (50, 360, 720)
    return c5_members(lo, hi, (nz, ny, nx)), total, 'strong', 'invert_omega %dx%dx%d, %d time steps (BASELINE configs[4])' % (nx, ny, nz, total)


def c5_members(lo, hi, shape=(50, 360, 720)):
    """Volumes lo..hi-1 of the 120-step omega batch.  Generated in FIXED blocks of eight steps (a 120-step forcing is
    12 GB per array on the host), block b always from seed + 8 b with eight steps, so that volume m holds the same
    numbers however the batch is split over ranks (the two-rank / one-rank flag comparison of the tests)."""
    from xinvert_amd import synthetic
    parts = []
    for b in range(lo // 8, (hi + 7) // 8):
        q = synthetic.omega_latlon(shape[0], shape[1], shape[2], steps=8, seed=synthetic.SEED + 8 * b)
        a, e = max(lo, 8 * b) - 8 * b, min(hi, 8 * b + 8) - 8 * b
        q['S0'] = q['S0'][a:e]
        q['coefs'] = [c if k in q['shared'] else c[a:e] for k, c in enumerate(q['coefs'])]
        parts.append(q)
    p = dict(parts[0])
    p['S0'] = np.concatenate([q['S0'] for q in parts])
    p['coefs'] = [c if k in p['shared'] else np.concatenate([q['coefs'][k] for q in parts])
                  for k, c in enumerate(p['coefs'])]
    return p


def time_resident(rp, sweeps, steps, warmup, **opts):
    """-> (wall seconds of `steps` solves, sum of HIP-event ms over the sweep launches, launches, last stats, flags)"""
    import torch
    for _ in range(warmup):
        rp.reset(); rp.solve(sweeps - 1, 0.0, **opts)
    rp.reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ms, nl = 0.0, 0
    for _ in range(steps):
        fl, s = rp.solve(sweeps - 1, 0.0, **opts)
        ms += s['sweep_ms']; nl += s['sweep_launches']
    torch.cuda.synchronize()
    return time.perf_counter() - t0, ms, nl, s, fl


def load_traffic():
    try:
        return json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))
    except Exception:
        return {}


_SRC_SHA = None


def src_sha():
    """hash of the library's sources in THIS tree (xinvert_amd/build.py: source_hash)"""
    global _SRC_SHA
    if _SRC_SHA is None:
        from xinvert_amd import build as xbuild
        _SRC_SHA = xbuild.source_hash()
    return _SRC_SHA


def traffic_entry(d, key):
    """A counter figure of profiles/traffic.json -- ONLY when the entry was profiled on the sources of this very tree (every
    entry carries the hash of xinvert_amd/csrc + include/xinv.h it was measured on): a kernel change that forgot to
    re-profile carries no stale bytes into the line.  -> (entry dict or None, why not)."""
    e = d.get(key)
    if e is None:
        return None, 'no entry'
    if not isinstance(e, dict):                              # (a bare number: its details sit beside it)
        det = d.get(key + '_detail') or {}
        e = dict(det, bytes_per_launch=det.get('bytes_per_launch', e))
    if e.get('src_sha') != src_sha():
        return None, 'stale: profiled on sources %s, this tree is %s' % (e.get('src_sha'), src_sha())
    return e, None


def alg_figures(kind, pts_per_launch, avg_ms):
    """SURVEY 8(d)'s comparable number: the reference kernel's operand set (48 / 72 B per point-sweep) over the
    launch time.  Labelled, because it is a fraction of the HBM roof only for a variant that streams all of it."""
    g = ALG_BYTES[kind] * pts_per_launch / (avg_ms * 1e-3) / 1e9
    return {'alg_bytes_per_point_sweep': ALG_BYTES[kind], 'alg_GBps': g, 'alg_frac': g / HBM_PEAK_GBS,
            'alg_frac_note': 'SURVEY 8(d) bytes x point-sweeps of the launch / launch time / 8 TB/s: a comparable figure, '
                             'NOT an HBM fraction when > 1 (fused sweeps, per-row coefficients, skipped tiles, Infinity Cache)'}


def config_lines(local):
    """The other BASELINE configurations on this GPU at SURVEY 8(d)'s sweep counts and one GPU's share of the
    batch (of eight): a timed run + an oracle parity check each."""
    import oracle as orc
    from xinvert_amd import synthetic
    from xinvert_amd.resident import ResidentProblem
    todo = [
        ('C1', 'invert_Poisson 360x180 lat-lon, one slice (BASELINE configs[0])',
         lambda: synthetic.poisson_latlon(180, 360, mask=False), 500, 10, orc.COLOUR_2, 25),
        ('C3-Stommel', 'invert_Stommel 2000x2000 Cartesian, R(x,y) varying (BASELINE configs[2])',
         lambda: synthetic.stommel_cartesian(2000, 2000), 500, 5, orc.COLOUR_2, 12),
        ('C3-Munk', 'invert_StommelMunk 2000x2000 Cartesian, biharmonic form (BASELINE configs[2])',
         lambda: synthetic.munk_cartesian(2000, 2000), 500, 3, orc.COLOUR_AUTO, 10),
        ('C3-Munk-xy', 'invert_StommelMunk 2000x2000 Cartesian, biharmonic form with A4(x,y) and R(x,y) varying along both axes '
                       '(BASELINE configs[2]: "spatially-varying" coefficients)',
         lambda: synthetic.munk_cartesian(2000, 2000, varying=True), 500, 3, orc.COLOUR_AUTO, 10),
        ('C4', 'invert_GillMatsuno 1440x720, 8 of the 64 forcing members = one GPU\'s share of eight (BASELINE configs[3]; '
               '--config c4 runs all 64)',
         lambda: synthetic.gill_matsuno(720, 1440, 8), 500, 5, orc.COLOUR_2, 12),
        ('C5', 'invert_omega 720x360x50, 15 of the 120 time steps = one GPU\'s share of eight (BASELINE configs[4]; '
               '--config c5 runs all 120)',
         lambda: c5_members(0, 15), 200, 3, orc.COLOUR_2, 10),
        # not BASELINE configurations: the headline's and the omega grid one column wider -- periodic x with an ODD number of
        # columns (the even-ring layout of the streaming kernels), beside their even twins above
        ('C2-odd', 'invert_Poisson 3601x1800 (configs[1] one column wider: the odd-xc periodic seam), one slice',
         lambda: synthetic.poisson_latlon(1800, 3601, mask=True), 500, 5, orc.COLOUR_2, 22),
        ('C5-odd', 'invert_omega 721x360x50 (configs[4] one column wider), 15 volumes',
         lambda: synthetic.omega_latlon(50, 360, 721, steps=15), 200, 3, orc.COLOUR_2, 11),
    ]
    out = []
    traffic = load_traffic().get('configs', {})
    for name, wl, make, sweeps, steps, order, psw in todo:
        t_all = time.perf_counter()
        p = make()
        kind = p['kind']
        rp = ResidentProblem(p, device=local)
        # Three timed runs of `steps` steps; ALL are in the line (`values`, `run_launch_us`) and the MEDIAN stands.  (Round 5
        # kept the better of two because of one stalled run seen once -- a third of the usual rate on one line, never seen
        # again in this round's runs: profiles/r06_bench_runs.txt; an outlier now shows in `values` instead of being dropped.)
        runs = [time_resident(rp, sweeps, steps, 1 if r_ == 0 else 0, timing=1) for r_ in range(3)]
        run_values = [float(rp.nb) * rp.n * sweeps * steps / t[0] for t in runs]
        run_launch_us = [t[1] / max(t[2], 1) * 1e3 for t in runs]
        dt, ms, nl, s, fl = sorted(runs, key=lambda t: t[0])[1]
        ok = bool((fl[:, 2] == sweeps - 1).all() and not fl[:, 0].any())
        n_all = float(rp.nb) * rp.n
        avg_ms = ms / max(nl, 1)
        spl_mean = float(sweeps) * steps / max(nl, 1)
        r = roofline_of(kind, s, n_all * spl_mean, avg_ms)
        rp.reset()
        flp, _ = rp.solve(psw - 1, 0.0)
        res = rp.result()
        m_chk = sorted({0, rp.nb - 1})
        par = [oracle_parity(synthetic.member(p, m), res[m], flp[m], psw, order) for m in m_chk]
        line = alg_figures(kind, n_all * spl_mean, avg_ms)
        # HBM-side bytes of this kernel on this workload (PMC passes, profiles/traffic.json: per point-sweep, static)
        tr, why = traffic_entry(traffic, name)
        if tr and tr.get('kernel_prefix') and kernel_name(kind, s).startswith(tr['kernel_prefix']) and tr.get('members') == rp.nb:
            tb = tr['bytes_per_point_sweep'] * n_all * spl_mean
            line.update({'traffic': tb, 'traffic_bytes_per_point_sweep': tr['bytes_per_point_sweep'],
                         'traffic_GBps': tb / (avg_ms * 1e-3) / 1e9,
                         'traffic_frac_of_hbm_peak': tb / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         'traffic_source': 'static: ' + tr.get('source', 'profiles/traffic.json')})
        else:
            line['traffic'] = None
            line['traffic_why_null'] = why or 'profiled kernel / member count differs from this run'
        out.append(line)
        line.update({'name': name, 'workload': wl, 'value': n_all * sweeps * steps / dt, 'unit': 'point-sweeps/s',
                    'values': run_values, 'value_is': 'median of the three timed runs in `values`',
                    'run_launch_us': run_launch_us, 'run_spread': max(run_values) / min(run_values),
                    'members': rp.nb, 'sweeps_per_step': sweeps, 'steps': steps, 'timed_runs': 3, 'ran_all_sweeps': ok,
                    'kernel': kernel_name(kind, s), 'sweeps_per_launch': s['sweeps_per_launch'],
                    'rows_per_tile': s['rows_per_tile'], 'lanes': s.get('lanes', 1),
                    'k_chunks': s.get('k_chunks', 0), 'cut_tiles': s.get('cut_tiles', 0), 'bound': r['bound'], 'frac': r['frac'], 'achieved': r['achieved'],
                    'peak': r['peak'], 'unit_roofline': r['unit'], 'valu_frac': r['valu_frac'],
                    'streamed_frac_of_hbm_peak': r['streamed_frac_of_hbm_peak'],
                    'streamed_bytes_per_point_sweep': r['streamed_bytes_per_point_sweep'],
                    'avg_launch_us': avg_ms * 1e3,
                    'parity_bitwise': bool(all(q['bitwise'] and q['loop_equal'] for q in par)),
                    'parity_sweeps': psw, 'parity_members': m_chk,
                    'seconds': time.perf_counter() - t_all})
        del rp
    return out


def sustained_leg(rp, sweeps, opts, n_all, seconds=5.0):
    """The headline solve back to back for >= `seconds` of wall clock: point-sweeps/s of every whole second and the mean
    launch duration in the first and the last of them -- what the fp64 clocks do under sustained load (the timed region
    of the contract line is twenty solves, under 0.1 s)."""
    import torch
    rp.reset(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = []                                               # (end time, sweep ms, launches) of every solve
    while True:
        fl, st = rp.solve(sweeps - 1, 0.0, **opts)
        torch.cuda.synchronize()
        marks.append((time.perf_counter() - t0, st['sweep_ms'], st['sweep_launches']))
        if marks[-1][0] >= seconds:
            break
    total = marks[-1][0]
    nsec = int(total)
    rates, launch = [], []
    for k in range(nsec):
        w = [m for m in marks if k <= m[0] < k + 1]
        rates.append(len(w) * n_all * sweeps / 1.0)
        launch.append(sum(m[1] for m in w) / max(1, sum(m[2] for m in w)))
    rs = sorted(rates)
    return {'seconds': total, 'solves': len(marks), 'value': len(marks) * n_all * sweeps / total, 'unit': 'point-sweeps/s',
            'per_second': rates, 'per_second_min': rs[0], 'per_second_median': rs[len(rs) // 2], 'per_second_max': rs[-1],
            'avg_launch_ms_first_second': launch[0], 'avg_launch_ms_last_second': launch[-1],
            'launch_drift': launch[-1] / launch[0] - 1.0,
            'note': 'every solve is followed by a host synchronisation (the per-second counts need its end time): '
                    'back-to-back solves, not one queue of launches'}


def host_pointer_solve(p, sweeps, reps=3, f32=False, **opt):
    """One C-ABI call on HOST arrays (the `_batched` entry): upload, sweeps, download -- what a caller of the reference's
    API pays.  -> (best wall seconds, stats of that call, S, flags).  f32: the first guess / solution and the forcing
    travel as float32 (xinv_options.f32_mask; promoted / rounded on the device) -- the dtype of every dataset the
    reference ships (tests/test_Poisson.py:14-24)."""
    import ctypes
    from xinvert_amd import _lib
    from xinvert_amd.resident import FN, scalars
    L = _lib.require_gpu()
    nb = p['S0'].shape[0]
    n = int(np.prod(p['S0'].shape[1:]))
    arrs, strides = [np.ascontiguousarray(p['S0'], dtype=np.float64)], [n]
    rowconst = 0
    for k, c in enumerate(p['coefs']):
        c = np.asarray(c)
        if k == 1 and p['kind'] in ('std2d', 'gen2d') and not c.any():
            arrs.append(None); strides.append(0)
        elif k in p['shared'] and k < len(p['coefs']) - 1 and c.strides[-1] == 0 and c.shape[-1] > 1:
            # a function of latitude handed over as the front end hands it over (core._prep_coef): one value per row
            arrs.append(np.ascontiguousarray(c[..., 0], dtype=np.float64)); strides.append(0); rowconst |= 1 << k
        else:
            arrs.append(np.ascontiguousarray(c, dtype=np.float64)); strides.append(0 if k in p['shared'] else n)
    q = {k: v for k, v in p.items() if k not in ('S0', 'coefs')}
    f32_mask = 0
    if f32:
        arrs[0] = arrs[0].astype(np.float32)
        arrs[-1] = arrs[-1].astype(np.float32)
        f32_mask = 1 | (1 << (len(arrs) - 1))
    isf = [bool((f32_mask >> k) & 1) for k in range(len(arrs))]
    best = None
    for _ in range(reps):
        S = arrs[0].copy()
        fl = np.tile(np.array([0., 1., 0.]), (nb, 1))
        o = _lib.options(rowconst_mask=rowconst, f32_mask=f32_mask, **opt)
        t = time.perf_counter()
        rc = getattr(L, FN[p['kind']] + '_batched')(_lib.hptr(S, f32=isf[0]), *[_lib.hptr(x, f32=isf[k + 1]) for k, x in enumerate(arrs[1:])], nb,
                                                    _lib.strides_arg(strides), *scalars(q), _lib.hptr(fl),
                                                    sweeps - 1, 0.0, ctypes.byref(o))
        dt = time.perf_counter() - t
        _lib.check(rc)
        if best is None or dt < best[0]:
            best = (dt, _lib.last_stats(), S, fl)
    return best


def end_to_end_leg(local, p_c2, S_resident):
    """Host arrays in, host arrays out -- what a caller of the reference's API sees (core.py:129-139, apps.py:1324-1394):
    `apps.invert_Poisson` on the headline slice, and the C-ABI host-pointer entry on one GPU's share of C5 and C4.
    Each with the wall clock, the upload / sweep / download spans, and the device-resident time of the same solve."""
    import xinvert_amd as xa
    from xinvert_amd import synthetic
    from xinvert_amd.resident import ResidentProblem
    out = {'note': 'PCIe-inclusive: never the headline `value` (inputs resident).  h2d_ms / d2h_ms are the spans of the copy '
                   'streams (they overlap the sweeps when the batch is cut into chunks)'}
    # ---- the front end on the headline slice
    ny, nx = p_c2['yc'], p_c2['xc']
    F = xa.Field(p_c2['zeta'][0], ('lat', 'lon'), {'lat': p_c2['lat'], 'lon': p_c2['lon']})
    iP = {'BCs': [p_c2['BCy'], p_c2['BCx']], 'mxLoop': 499, 'tolerance': 0.0, 'printInfo': False, 'device_prep': True,
          'device': local}
    xa.invert_Poisson(F, ['lat', 'lon'], iParams=iP)          # (library load, pools, pinned rings)
    best, S = None, None
    for _ in range(5):
        t = time.perf_counter()
        S = xa.invert_Poisson(F, ['lat', 'lon'], iParams=iP)
        dt = time.perf_counter() - t
        if best is None or dt < best[0]:
            best = (dt, S.iParams['stats'])
    dt, st = best
    sea = ~np.isnan(p_c2['zeta'][0])
    out['C2_invert_Poisson'] = {
        'workload': 'apps.invert_Poisson(zeta [%d, %d] float64 with NaN land, BCs fixed / periodic, 500 sweeps): mask, '
                    'cos(lat) scaling and de-mask on the device, result as a host array' % (ny, nx),
        'wall_ms': dt * 1e3, 'library_call_ms': st['wall_ms'], 'h2d_ms': st['h2d_ms'], 'd2h_ms': st['d2h_ms'],
        'python_ms': dt * 1e3 - st['wall_ms'], 'value_pcie_inclusive': float(ny) * nx * 500 / dt,
        'bitwise_equal_to_resident': bool(S_resident is not None and np.array_equal(S.values[sea], S_resident[sea]))}
    # ---- one GPU's share of the batched configurations through host pointers
    for name, make, sweeps in (('C5x15', lambda: c5_members(0, 15), 200),
                               ('C4x8', lambda: synthetic.gill_matsuno(720, 1440, 8), 500)):
        p = make()
        rp = ResidentProblem(p, device=local)
        tr = min(time_resident(rp, sweeps, 1, 1 if r_ == 0 else 0)[0] for r_ in range(3))
        # ... and the same resident solve on a GPU that has been idle for 50 ms, as it is when a host-pointer call arrives: the
        # clocks fall within milliseconds of idling and take tens of milliseconds of load to come back (tools/idle_clock_probe.py)
        import torch
        rp.reset(); torch.cuda.synchronize(); time.sleep(0.05)
        tc0 = time.perf_counter(); rp.solve(sweeps - 1, 0.0); torch.cuda.synchronize(); tr_cold = time.perf_counter() - tc0
        rp.reset(); rp.solve(sweeps - 1, 0.0)
        res = rp.result()
        del rp
        dt, st, S, fl = host_pointer_solve(p, sweeps, device=local)
        n_all = float(np.prod(p['S0'].shape))
        out[name] = {'wall_ms': dt * 1e3, 'h2d_ms': st['h2d_ms'], 'd2h_ms': st['d2h_ms'], 'sweep_ms_resident': tr * 1e3,
                     'sweep_ms_resident_after_50ms_idle': tr_cold * 1e3,
                     'host_chunks': st['host_chunks'], 'vs_resident': dt / tr,
                     'value_pcie_inclusive': n_all * sweeps / dt, 'unit': 'point-sweeps/s',
                     'bitwise_equal_to_resident': bool(np.array_equal(S, res))}
        if name == 'C5x15':
            # ... and as the reference's users hold their data: float32 forcing in, float32 solution out (half the bytes over
            # PCIe; the sweeps are the same float64 arithmetic on the promoted values).  Checked against the resident solve of
            # the float32-rounded forcing, rounded to float32.
            p32 = dict(p)
            p32['coefs'] = list(p['coefs'][:-1]) + [np.asarray(p['coefs'][-1]).astype(np.float32).astype(np.float64)]
            p32['S0'] = np.asarray(p['S0']).astype(np.float32).astype(np.float64)
            rp = ResidentProblem(p32, device=local)
            rp.reset(); rp.solve(sweeps - 1, 0.0)
            res32 = rp.result().astype(np.float32)
            del rp
            dt32, st32, S32, _ = host_pointer_solve(p32, sweeps, device=local, f32=True)
            out[name + '_float32'] = {'wall_ms': dt32 * 1e3, 'h2d_ms': st32['h2d_ms'], 'd2h_ms': st32['d2h_ms'],
                                      'host_chunks': st32['host_chunks'], 'vs_resident': dt32 / tr,
                                      'value_pcie_inclusive': n_all * sweeps / dt32, 'unit': 'point-sweeps/s',
                                      'note': 'first guess / solution and forcing as float32 host arrays (f32_mask), float64 sweeps',
                                      'equal_to_resident_rounded_to_float32': bool(S32.dtype == np.float32 and np.array_equal(S32, res32))}
    return out


def inproc_leg(a, ngpu):
    """The in-process multi-device mode: host pointers, one call, xinv_options.ndev = ngpu."""
    import ctypes
    from xinvert_amd import _lib, synthetic
    from xinvert_amd.resident import FN, scalars
    L = _lib.require_gpu()
    nb = max(ngpu, a.members or ngpu) if a.config == 'c2' else (a.members or 8 * ngpu)
    p = synthetic.poisson_latlon(a.ny, a.nx, mask=True, members=nb) if a.config == 'c2' else \
        (synthetic.gill_matsuno(720, 1440, nb) if a.config == 'c4' else synthetic.omega_latlon(50, 360, 720, steps=nb))
    sweeps = a.sweeps or (200 if a.config == 'c5' else 500)
    n = int(np.prod(p['S0'].shape[1:]))
    arrs, strides = [np.ascontiguousarray(p['S0'], dtype=np.float64)], [n]
    for k, c in enumerate(p['coefs']):
        if k == 1 and p['kind'] in ('std2d', 'gen2d') and not np.asarray(c).any():
            arrs.append(None); strides.append(0)
        else:
            arrs.append(np.ascontiguousarray(c, dtype=np.float64)); strides.append(0 if k in p['shared'] else n)
    fl = np.tile(np.array([0., 1., 0.]), (nb, 1))
    o = _lib.options(devices=list(range(ngpu)), timing=1)
    q = {k: v for k, v in p.items() if k not in ('S0', 'coefs')}
    best = None
    for _ in range(3):
        S = arrs[0].copy()
        t = time.perf_counter()
        rc = getattr(L, FN[p['kind']] + '_batched')(_lib.hptr(S), *[_lib.hptr(x) for x in arrs[1:]], nb,
                                                    _lib.strides_arg(strides), *scalars(q), _lib.hptr(fl),
                                                    sweeps - 1, 0.0, ctypes.byref(o))
        dt = time.perf_counter() - t
        _lib.check(rc)
        best = dt if best is None else min(best, dt)
    st = _lib.last_stats()
    return {'mode': 'one process, host pointers, xinv_options.ndev = %d (one host thread per GPU, no collective)' % ngpu,
            'devices_used': st['devices'], 'members': nb, 'sweeps': sweeps,
            'value_pcie_inclusive': float(nb) * n * sweeps / best, 'unit': 'point-sweeps/s',
            'wall_ms': best * 1e3, 'h2d_ms': st['h2d_ms'], 'd2h_ms': st['d2h_ms'], 'sweep_ms': st['sweep_ms'],
            'note': 'upload + sweeps + download of host arrays: never the headline `value`'}


def self_launch(a):
    """--gpus N > 1 outside torchrun: start the N ranks ourselves and relay their exit status."""
    import torch
    forced = 'XINV_FORCE_DEVICE' in os.environ              # testing aid: several ranks share one GPU (gloo)
    ndev = torch.cuda.device_count()
    if ndev < a.gpus and not forced:
        raise SystemExit('bench.py: --gpus %d but only %d GPU(s) visible; refusing to run fewer ranks than asked '
                         '(set XINV_FORCE_DEVICE=0 XINV_DIST_BACKEND=gloo to share one GPU in a test)' % (a.gpus, ndev))
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0)); port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(a.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(a))
    import torch
    from xinvert_amd import _lib, synthetic
    from xinvert_amd import dist as xdist
    from xinvert_amd.resident import ResidentProblem

    rank, local, world = xdist.init_process_group()
    joined = torch.distributed.is_available() and torch.distributed.is_initialized()
    if world != a.gpus:
        raise SystemExit('WORLD_SIZE %d != --gpus %d' % (world, a.gpus))
    _lib.require_gpu()
    # XINV_FORCE_DEVICE: testing aid (several ranks on one GPU with the gloo backend)
    local = int(os.environ.get('XINV_FORCE_DEVICE', local))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    sweeps = a.sweeps or (200 if a.config == 'c5' else 500)

    p, total_members, scaling, wl_name = build_problem(a, rank, world)
    kind = p['kind']
    rp = ResidentProblem(p, device=local)          # inputs resident in HBM before the timed region
    nb, n = rp.nb, rp.n
    opts = dict(sweeps_per_launch=a.spl, rows_per_tile=a.rows, timing=1)

    def step():
        fl, s = rp.solve(sweeps - 1, 0.0, **opts)
        allf = xdist.gather_flags(fl, total_members) if joined else fl
        return s, allf

    def barrier():
        if joined:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # library initialisation (code-object load, events, detection buffers) on a toy problem: set-up,
    # like creating the tensors above -- not a warm-up step of the workload
    toy = ResidentProblem(synthetic.poisson_latlon(8, 16, mask=False, seed=1), device=local)
    toy.solve(1, 0.0)
    torch.cuda.synchronize()

    for _ in range(a.warmup):
        rp.reset()
        step()
    import gc
    gc.disable()                                   # (as timeit does: no collector pause inside the timed region; no gc.collect()
    rp.reset()                                     #  here -- 20 ms of idle GPU before t0 cost the first steps 10 %: the clocks
    barrier()                                      #  ramp for ~30 ms after any idle gap, tools/step_time_vs_state.py)
    t0 = time.perf_counter()
    ms_sweeps, launches = 0.0, 0
    stamps = [t0]                                  # (host clock after every step: a stall inside the timed region shows in the line)
    for _ in range(a.steps):
        s, allf = step()
        ms_sweeps += s['sweep_ms']; launches += s['sweep_launches']
        stamps.append(time.perf_counter())
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0              # this rank's own K steps (before it waits for the others)
    gc.enable()
    step_ms_order = [(b_ - a_) * 1e3 for a_, b_ in zip(stamps[:-1], stamps[1:])]
    step_ms = sorted(step_ms_order)
    barrier()
    dt = time.perf_counter() - t0
    ranks_done, backend = 1, None
    rank_values, n1 = None, None
    # a checksum of checksums over the FINAL state of every member: the wrap-around int64 sum of the bit patterns of its
    # S (order-independent, exact, one pass on the device), gathered over the ranks like the flags -- equal digests = the
    # same fields bit for bit however the batch was split.  (flags[:, 1], the relative change of mean|S|, is a sum whose
    # grouping follows the launch's tiling -- 1e-12 between two splits of a batch, DESIGN 5.3 -- and is not hashed.)
    ck = rp.S.view(torch.int64).reshape(nb, -1).sum(dim=1).cpu().numpy().reshape(nb, 1)
    all_ck = xdist.gather_blocks(ck, total_members) if joined else ck
    if joined:
        backend = torch.distributed.get_backend()
        tdev = dev if backend == 'nccl' else torch.device('cpu')
        tt = torch.tensor([dt], dtype=torch.float64, device=tdev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
        one = torch.ones(1, dtype=torch.float64, device=tdev)          # ranks that really took part
        torch.distributed.all_reduce(one, op=torch.distributed.ReduceOp.SUM)
        ranks_done = int(round(float(one.item())))
        # every rank's own rate on its own block (its K steps, its clock): a straggler GPU shows here
        mine = torch.tensor([float(nb) * n * sweeps * a.steps / dt_own, float(nb)], dtype=torch.float64, device=tdev)
        every = [torch.empty_like(mine) for _ in range(world)]
        torch.distributed.all_gather(every, mine)
        rank_values = [{'rank': r, 'members': int(round(float(e[1].item()))), 'value': float(e[0].item())}
                       for r, e in enumerate(every)]
        # rank 0 ALONE on its share, the others idle at the barrier: what N = 1 gives for the same per-GPU work
        # (weak scaling: directly comparable with the N = 1 run; strong scaling: x world = the linear-scaling line)
        if rank == 0:
            rp.reset()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(a.steps):
                rp.solve(sweeps - 1, 0.0, **opts)
            torch.cuda.synchronize()
            n1 = float(nb) * n * sweeps * a.steps / (time.perf_counter() - t1)
        barrier()
    assert (allf[:, 2] == sweeps - 1).all() and not allf[:, 0].any(), allf[:4]
    assert allf.shape[0] == total_members

    if rank == 0:
        total_ps = float(total_members) * n * sweeps * a.steps
        spl = s['sweeps_per_launch']
        avg_ms = ms_sweeps / max(launches, 1)
        # the timed launches: full K-sweep passes plus one shorter tail pass per step when K does
        # not divide the sweep count; per-launch figures below use the mean over all of them
        sweeps_per_launch_mean = float(sweeps) * a.steps / max(launches, 1)
        pts_per_launch = float(nb) * n * sweeps_per_launch_mean
        active = 1.0 - s['masked_tile_ppm'] / 1e6
        out = {
            'metric': 'SOR grid-points*iters/sec (fp64) at %dx%d%s, masked points counted'
                      % ((a.nx, a.ny, '') if a.config == 'c2' else
                         ((1440, 720, ' x %d members' % total_members) if a.config == 'c4' else
                          (720, 360, 'x50 x %d steps' % total_members))),
            'value': total_ps / dt, 'unit': 'point-sweeps/s',
            'n_gpus': ranks_done, 'gpus_requested': a.gpus, 'rccl_ranks': ranks_done if backend == 'nccl' else 0,
            'collective_backend': backend, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': dt / a.steps * 1e3, 'higher_is_better': True, 'scaling': scaling,
            'step_ms_min_median_max': [step_ms[0], step_ms[len(step_ms) // 2], step_ms[-1]],      # rank 0's own steps (host clock)
            'step_ms': [round(x, 3) for x in step_ms_order[:64]],                                   # ... in order (the first 64)
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'rank_values': rank_values, 'n1_value': n1,
            'rank_spread': None if not rank_values else max(r['value'] for r in rank_values) / min(r['value'] for r in rank_values),
            'linear_scaling_value': None if n1 is None else (n1 * ranks_done if scaling == 'weak' else n1 * ranks_done),
            'n1_value_note': None if n1 is None else 'rank 0 alone on its own block (%d member(s)), the other ranks idle: the '
                             'N = 1 rate for the same per-GPU work; linear scaling = n1_value x n_gpus' % nb,
            'flags_sha256': __import__('hashlib').sha256(np.ascontiguousarray(allf[:, [0, 2]], dtype=np.float64).tobytes()).hexdigest(),
            'flags_sha256_of': 'overflow flag and loop index of every slice, gathered over the ranks',
            'S_checksum_sha256': __import__('hashlib').sha256(np.ascontiguousarray(all_ck, dtype=np.int64).tobytes()).hexdigest(),
            'S_checksum_of': 'per slice: wrap-around int64 sum of the bit patterns of its final S; gathered over the ranks',
            'S_checksums_first': [int(v) for v in np.asarray(all_ck).reshape(-1)[:16]],
            'config': {'workload': wl_name, 'sweeps_per_step': sweeps,
                       'members_total': total_members, 'members_this_gpu': nb,
                       'sweeps_per_launch': spl, 'rows_per_tile': s['rows_per_tile'], 'lanes': s.get('lanes', 1),
                       'xuniform_mask': s['xuniform_mask'], 'masked_tile_pct': s['masked_tile_pct'],
                       'masked_tile_share': 1.0 - active,
                       'path': {1: 'colour', 2: 'fused'}.get(s['path'], '?'),
                       'parallelism': 'batch-axis shard x%d' % world},
        }
        out['value_active'] = out['value'] * active
        rr = roofline_of(kind, s, pts_per_launch, avg_ms)
        # The keys a reader needs to recompute the line come FIRST (the driver's record keeps the first 24 keys of this
        # object: VERDICT r4 weak 4); they are filled in below as the legs that produce them run.
        roof = {k: None for k in ROOF_FIRST}
        roof.update(rr)
        roof.update({
            'note': 'point updates of the tiles that RAN (skipped, fully masked tiles are not counted: active_tile_share) '
                    'x useful fp64 operations / mean launch duration; `streamed_*`: bytes this kernel variant must move',
            'peak_note': 'valu_fp64: fp64 vector issue rate without FMA (256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz; '
                         'contraction is off by the bit-exactness contract; datasheet FMA peak %.1f); hbm: 8 TB/s spec'
                         % FP64_FMA_SPEC_TFLOPS,
            'frac_of_measured_valu_peak': rr['valu_TFLOPs'] / FP64_VALU_MEASURED_TFLOPS,
            'frac_of_fma_datasheet_peak': rr['valu_TFLOPs'] / FP64_FMA_SPEC_TFLOPS,
            'useful_flops_per_point_update': UPD_FLOPS[kind],
            'kernel': kernel_name(kind, s),
            'launches': int(launches),
            # per step: wall clock minus the HIP-event time of the sweep launches -- what a solve spends before its first
            # and after its last launch (with a resident plan: control-block reset, the last pass's norm, the copy of S
            # out of the buffer the rotation ended in); VERDICT r4 item 2 asks for <= 0.08 ms
            'ms_outside_launches': (dt * 1e3 - ms_sweeps) / a.steps,
            'planned': int(s.get('planned', 0)),
            'alg_bytes_per_launch': ALG_BYTES[kind] * pts_per_launch})
        roof.update(alg_figures(kind, pts_per_launch, avg_ms))
        em = tile_model(s) if kind in ('std2d', 'gen2d') else None
        if em:
            roof['executed_over_useful'] = em
        traffic, traffic_why = None, 'no profile of this configuration'
        tfile = os.path.join(ROOT, 'profiles', 'traffic.json')
        tj = load_traffic()
        if a.config == 'c2' and (a.ny, a.nx) == (1800, 3600) and nb == 1 and a.mask == 'continents':
            e, traffic_why = traffic_entry(tj, ('std2d_pipe_um%d' % s['xuniform_mask']) if s.get('pipelined')
                                           else 'std2d_spl%d_um%d' % (spl, s['xuniform_mask']))
            traffic = e['bytes_per_launch'] if e else None
        elif a.config == 'c4' and nb == 64 and s.get('pipelined') and s['xuniform_mask'] == 31:
            e, traffic_why = traffic_entry(tj, 'gen2d_pipe_um31_fr_c4x64')     # (the 64-member launch profiled by tools/profile_headline.sh)
            traffic = e['bytes_per_launch'] if e else None
        elif a.config == 'c5' and spl == 2:
            e, traffic_why = traffic_entry(tj.get('configs', {}), 'C5')        # per point-sweep, from the 15-volume launch
            traffic = e['bytes_per_point_sweep'] * pts_per_launch if e else None
        roof['traffic'] = traffic
        roof['traffic_source'] = 'static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel variant (profiles/traffic.json), not read in this run; used only while the entry\'s source hash matches this tree (src_sha)'
        roof['src_sha'] = src_sha()
        if not traffic:
            roof['traffic_why_null'] = traffic_why
        if traffic:
            roof['traffic_GBps'] = traffic / (avg_ms * 1e-3) / 1e9
            roof['traffic_frac_of_hbm_peak'] = roof['traffic_GBps'] / HBM_PEAK_GBS
        out['roofline'] = roof

        single = (world == 1)
        S_res500 = None
        if a.config == 'c2' and single and not a.no_hbm:
            # ---- the HBM-bound variant on a working set far beyond the Infinity Cache -----------------
            hm = 8
            ph = c2_problem(a, 0, hm, shared=False)
            hb = ResidentProblem(ph, device=local)
            hopts = dict(sweeps_per_launch=1, timing=1, no_xuniform=1, no_tile_skip=1)
            hsw = 60
            th, hms, hl, sh, fl_h = time_resident(hb, hsw, 3, 1, **hopts)
            h_avg = hms / max(hl, 1)
            pts = float(hm) * n
            ws_bytes = pts * 8.0 * 5                         # S, S2 (ping-pong twin), A, C, F per member
            h40 = 40.0 * pts / (h_avg * 1e-3) / 1e9
            h48 = 48.0 * pts / (h_avg * 1e-3) / 1e9
            hb.reset()
            flq, _ = hb.solve(9, 0.0, **hopts)
            hpar = oracle_parity(synthetic.member(ph, hm - 1), hb.result()[hm - 1], flq[hm - 1], 10)
            out['roofline_hbm'] = {
                'bound': 'hbm', 'achieved': h40, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': h40 / HBM_PEAK_GBS,
                'frac_of_achievable_6300': h40 / 6300.0,
                'bytes_counted_per_point_sweep': 40,
                'alg48_GBps': h48, 'alg48_frac': h48 / HBM_PEAK_GBS,
                'variant': 'k_fused2d<FusedStd2D, K=1, x-uniform mask=0>, %d members each with its own A, C, F: one sweep per pass, '
                           'S (read + write) A C F streamed in full = 40 B per point (B is identically zero: detected once, never '
                           'read; SURVEY 8(d) counts it: the 48 B figure is alg48_*), no tile skipping' % hm,
                'working_set_bytes': ws_bytes, 'members': hm,
                'value': pts * hsw * 3 / th, 'unit_value': 'point-sweeps/s',
                'avg_launch_ms': h_avg, 'launches': int(hl), 'lanes': sh.get('lanes', 1),
                'avg_launch_note': 'HIP-event time of the sweep passes / passes; with two lanes a pass is two kernel launches '
                                   'on two streams (each over half the members), overlapping',
                'sweeps_per_launch': sh['sweeps_per_launch'], 'xuniform_mask': sh['xuniform_mask'],
                'masked_tile_pct': sh['masked_tile_pct'], 'parity_bitwise_10_sweeps': hpar['bitwise'] and hpar['loop_equal'],
                'traffic': None}
            # PMC bytes of this very variant, static, PER PASS over the 8 members: the profiled kernel launch covers
            # members / lanes of them (two launch chains: `lanes`), so bytes per pass = bytes per launch x lanes
            hd, hwhy = traffic_entry(load_traffic(), 'std2d_spl1_um0_all')
            if hd and hd.get('lanes') and (a.ny, a.nx) == (1800, 3600) and int(hd.get('members', hm)) == hm:
                ht = float(hd['bytes_per_launch']) * int(hd['lanes'])
                out['roofline_hbm'].update({'traffic': ht, 'traffic_GBps': ht / (h_avg * 1e-3) / 1e9,
                                            'traffic_frac_of_hbm_peak': ht / (h_avg * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                            'traffic_bytes_per_point_sweep': ht / pts,
                                            'traffic_lanes_profiled': int(hd['lanes']),
                                            'traffic_source': roof['traffic_source']})
            else:
                out['roofline_hbm']['traffic_why_null'] = hwhy or 'profile of another launch shape'
            # the one figure of this line that IS a fraction of the HBM roof, next to the headline kernel's own bound
            out['roofline']['hbm_variant'] = {k: out['roofline_hbm'][k] for k in
                                              ('bound', 'achieved', 'peak', 'unit', 'frac', 'frac_of_achievable_6300',
                                               'bytes_counted_per_point_sweep', 'alg48_frac', 'avg_launch_ms', 'members',
                                               'working_set_bytes', 'traffic', 'parity_bitwise_10_sweeps')}
            out['roofline']['hbm_variant'].update({k: out['roofline_hbm'][k] for k in ('traffic_GBps', 'traffic_frac_of_hbm_peak')
                                                   if k in out['roofline_hbm']})
            out['roofline']['hbm_frac'] = out['roofline_hbm']['frac']
            del hb

        if a.config == 'c2' and single and not a.no_parity:
            psw = a.parity_sweeps or sweeps
            rp.reset()
            fl_p, sp_ = rp.solve(psw - 1, 0.0, **opts)
            same_cfg = all(sp_[k] == s[k] for k in ('path', 'sweeps_per_launch', 'rows_per_tile',
                                                    'xuniform_mask', 'masked_tile_pct', 'pipelined'))
            S_res = rp.result()[0]
            S_res500 = S_res if psw == 500 else None
            par = oracle_parity(synthetic.member(p, 0), S_res, fl_p[0], psw)
            par['same_kernel_config_as_timed'] = bool(same_cfg)
            out['parity'] = par
            out['roofline']['parity_bitwise'] = bool(par['bitwise'] and par['loop_equal'] and same_cfg)
        if a.config == 'c2' and single and not a.no_configs:
            out['sustained'] = sustained_leg(rp, sweeps, opts, float(nb) * n, a.sustained_seconds)
            out['roofline']['sustained_value'] = out['sustained']['value']
            out['roofline']['sustained_launch_drift'] = out['sustained']['launch_drift']
        if a.config == 'c2' and single and not a.no_configs:
            # the same solve WITHOUT a resident plan (every call of xinv_<form>_f64_dev re-derives what a plan keeps: what
            # rounds 1-4 timed) -- for continuity with their lines
            rq = ResidentProblem(p, device=local, plan=False)
            tq, mq, lq, sq, _ = time_resident(rq, sweeps, 5, 2, timing=1)
            out['unplanned'] = {'value': float(nb) * n * sweeps * 5 / tq, 'unit': 'point-sweeps/s', 'ms_per_step': tq / 5 * 1e3,
                                'avg_launch_us': mq / max(lq, 1) * 1e3, 'planned': int(sq.get('planned', 0)),
                                'note': 'ResidentProblem(plan=False): detection passes, per-row records and tile lists rebuilt in every solve'}
            del rq
        if a.config == 'c2' and single and a.mask == 'continents' and not a.no_configs:
            # mask sensitivity of the headline: the same solve with a coastline-scale mask
            # (few whole tiles to skip) and with tile skipping off
            alt = {}
            for tag, mk, o2 in (('coastline_mask', 'coastline', {}), ('no_tile_skip', None, {'no_tile_skip': 1})):
                p2 = synthetic.poisson_latlon(a.ny, a.nx, mask=mk, members=nb) if mk else p
                r2 = ResidentProblem(p2, device=local) if mk else rp
                t2, m2, l2, s2, _ = time_resident(r2, sweeps, 5, 1, timing=1, **o2)
                alt[tag] = {'value': float(nb) * n * sweeps * 5 / t2, 'masked_tile_share': s2['masked_tile_ppm'] / 1e6,
                            'land_share': float(np.mean(p2['coefs'][3] == p2['undef'])),
                            'avg_launch_us': m2 / max(l2, 1) * 1e3, 'kernel': kernel_name(kind, s2)}
            out['mask_sensitivity'] = alt
        if a.config == 'c2' and single and not a.no_configs:
            # ---- the opt-in contracted arithmetic (XINV_FLAG_FMA) on the same workload: NOT the headline (the default
            # path is the reference's arithmetic bit for bit); checked bit for bit against the oracle's XO_FMA restatement
            import oracle as orc
            tf, mf, lf, sf, _ = time_resident(rp, sweeps, 5, 1, timing=1, fma=1)
            fsw = min(sweeps, 100)
            rp.reset()
            fl_f, _ = rp.solve(fsw - 1, 0.0, fma=1)
            fpar = oracle_parity(synthetic.member(p, 0), rp.result()[0], fl_f[0], fsw, orc.COLOUR_2 | orc.FMA)
            out['contracted'] = {'flag': 'XINV_FLAG_FMA (opt-in; explicit fma at fixed positions of the update)',
                                 'value': float(nb) * n * sweeps * 5 / tf, 'unit': 'point-sweeps/s',
                                 'vs_default': float(nb) * n * sweeps * 5 / tf / out['value'],
                                 'avg_launch_us': mf / max(lf, 1) * 1e3, 'kernel': kernel_name(kind, sf),
                                 'parity_vs_fma_oracle': fpar,
                                 'tied_to_reference': 'tests/test_fma_oracle.py: <= 1e-12 relative to the reference\'s golden '
                                                      'vectors after their sweeps, <= 1e-6 rel-L2 converged'}
        del rp
        if a.config == 'c2' and single and not a.no_configs:
            out['configs'] = config_lines(local)
            # (the driver's record keeps the contract keys: a compact copy of the table rides inside `roofline`)
            out['roofline']['configs'] = [{k: c.get(k) for k in ('name', 'value', 'members', 'sweeps_per_step', 'kernel', 'bound',
                                                                  'frac', 'valu_frac', 'streamed_frac_of_hbm_peak', 'alg_frac',
                                                                  'traffic_bytes_per_point_sweep', 'traffic_frac_of_hbm_peak',
                                                                  'parity_bitwise')} for c in out['configs']]
        if a.config == 'c2' and single and not a.no_e2e and (a.ny, a.nx) == (1800, 3600) and a.mask == 'continents':
            out['end_to_end'] = end_to_end_leg(local, p, S_res500)
            e2 = out['end_to_end']
            out['roofline'].update({'e2e_c2_invert_poisson_ms': e2['C2_invert_Poisson']['wall_ms'], 'e2e_c5x15_ms': e2['C5x15']['wall_ms'],
                                    'e2e_c5x15_vs_resident': e2['C5x15']['vs_resident'], 'e2e_c4x8_ms': e2['C4x8']['wall_ms']})
        if a.inproc and single:
            out['inproc'] = inproc_leg(a, a.gpus)
        if a.config == 'c2' and single and not a.no_cpu:
            out['cpu_baseline'] = cpu_baseline(synthetic.member(p, 0), a.cpu_seconds)
        print(json.dumps(out))
    if joined:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
