#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X SOR inversion engine.

Metric (BASELINE.json): SOR grid-points x iterations per second, fp64, on the 3600 x 1800
global lat-lon Poisson problem with a land/sea mask (configs[1]).  One *step* is one complete
hot-path pass: `xinv_standard_2d_f64_dev` over one batch of synthetic input already resident in
HBM, running a fixed number of sweeps (tolerance = 0, mxLoop = sweeps - 1), norm + stopping
rule evaluated on the device after every sweep exactly as in production.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--sweeps S] [--spl 1..4] [--members M]

N > 1: launched by torch.distributed.run, one rank per GPU; every rank solves its own
member(s) of the batch axis (weak scaling, no data-path collective) and the per-slice flags
are all-gathered over RCCL after each step.  Rank 0 prints ONE JSON line.

Extra objects in the JSON line:
  roofline      achieved = algorithmic bytes per sweep launch (48 B x grid points x sweeps per
                launch) / mean launch duration, measured live with HIP events
                on the solve's stream over the timed region (xinv_stats.sweep_ms).
  cpu_baseline  the lexicographic C restatement of the reference (oracle/, 1 core) timed on
                this box's host on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES = {'std2d': 48, 'gen2d': 72, 'std3d': 48}
HBM_PEAK_GBS = 8000.0                                       # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--sweeps', type=int, default=500, help='SOR sweeps per step (SURVEY.md 8(d): 500 for C1-C4)')
    ap.add_argument('--spl', type=int, default=0, help='sweeps fused per launch (0 = engine default)')
    ap.add_argument('--rows', type=int, default=0, help='rows per tile (0 = engine default)')
    ap.add_argument('--members', type=int, default=1, help='batch members per GPU')
    ap.add_argument('--ny', type=int, default=1800)
    ap.add_argument('--nx', type=int, default=3600)
    ap.add_argument('--no-cpu', action='store_true', help='skip the cpu_baseline leg')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    return ap.parse_args()


def cpu_baseline(p, budget_s):
    """Time the oracle's lexicographic sweep (the reference's execution model: one slice, one
    core) on member 0 of the same workload for about `budget_s` seconds."""
    import oracle as orc
    from xinvert_amd import synthetic
    orc.build()
    q = synthetic.member(p, 0)
    c = [np.ascontiguousarray(a, dtype=np.float64) for a in q['coefs']]
    npts = q['yc'] * q['xc']

    def run(nsweeps):
        S = np.array(q['S0'], dtype=np.float64, copy=True)
        fl = np.array([0., 1., 0.])
        t = time.perf_counter()
        orc.standard_2d(S, *c, q['yc'], q['xc'], q['dely'], q['delx'], q['BCy'], q['BCx'],
                        q['delxSqr'], q['ratioQtr'], q['ratioSqr'], q['optArg'], q['undef'], fl,
                        nsweeps - 1, 0.0, orc.LEX)
        return time.perf_counter() - t

    t2 = run(2)
    n = int(max(4, min(400, budget_s / max(t2 / 2, 1e-6))))
    t = run(n)
    out = {'value': npts * n / t, 'unit': 'point-sweeps/s', 'cores': 1, 'kind': 'port',
           'sample': '%d lexicographic sweeps of the full %dx%d slice (%.1f s), oracle/xinv_oracle.c '
                     'gcc -O3 -ffp-contract=off' % (n, q['yc'], q['xc'], t)}
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
    return out


def main():
    a = parse()
    import torch
    from xinvert_amd import _lib, synthetic
    from xinvert_amd import dist as xdist

    rank, local, world = xdist.init_process_group()
    joined = torch.distributed.is_available() and torch.distributed.is_initialized()
    if world != a.gpus and world > 1:
        raise SystemExit('WORLD_SIZE %d != --gpus %d' % (world, a.gpus))
    L = _lib.require_gpu()
    # XINV_FORCE_DEVICE: testing aid (several ranks on one GPU with the gloo backend)
    local = int(os.environ.get('XINV_FORCE_DEVICE', local))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    # synthetic workload: same grid on every rank, rank-dependent seed (independent members)
    p = synthetic.poisson_latlon(a.ny, a.nx, mask=True, seed=synthetic.SEED + rank, members=a.members)
    n = a.ny * a.nx
    nb = a.members
    S0 = torch.from_numpy(np.ascontiguousarray(p['S0'])).to(dev)
    S = S0.clone()
    coefs = [torch.from_numpy(np.ascontiguousarray(c, dtype=np.float64)).to(dev) for c in p['coefs']]
    # the cross coefficient B of invert_Poisson is identically zero: it travels as NULL, exactly as
    # the front end (xinvert_amd/core.py:_prep_coef) hands it to the library
    b_null = not np.asarray(p['coefs'][1]).any()
    strides = [n] + [0 if k in p['shared'] else n for k in range(len(coefs))]
    st = _lib.strides_arg(strides)
    flags = np.tile(np.array([0., 1., 0.]), (nb, 1))
    opt = _lib.options(device=local, sweeps_per_launch=a.spl, rows_per_tile=a.rows, timing=1)
    stream = torch.cuda.current_stream()
    sp = ctypes.c_void_p(stream.cuda_stream)
    b = _lib.bc

    def step():
        rc = L.xinv_standard_2d_f64_dev(
            ctypes.c_void_p(S.data_ptr()),
            *[None if (k == 1 and b_null) else ctypes.c_void_p(c.data_ptr()) for k, c in enumerate(coefs)],
            nb, st, a.ny, a.nx, p['dely'], p['delx'], b(p['BCy']), b(p['BCx']), p['delxSqr'],
            p['ratioQtr'], p['ratioSqr'], p['optArg'], p['undef'], _lib.hptr(flags),
            a.sweeps - 1, 0.0, ctypes.byref(opt), sp)
        _lib.check(rc)
        s = _lib.last_stats()
        allf = xdist.gather_flags(flags, nb * world) if joined else flags
        return s, allf

    def barrier():
        if joined:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # library initialisation (code-object load, events, detection buffers) on a toy problem: set-up,
    # like creating the tensors above -- not a warm-up step of the workload
    _t = synthetic.poisson_latlon(8, 16, mask=False, seed=1)
    _ts = torch.from_numpy(np.ascontiguousarray(_t['S0'])).to(dev)
    _tc = [torch.from_numpy(np.ascontiguousarray(c, dtype=np.float64)).to(dev) for c in _t['coefs']]
    _tf = np.array([[0., 1., 0.]])
    _lib.check(L.xinv_standard_2d_f64_dev(
        ctypes.c_void_p(_ts.data_ptr()), *[ctypes.c_void_p(c.data_ptr()) for c in _tc], 1,
        _lib.strides_arg([128, 0, 0, 0, 128]), 8, 16, _t['dely'], _t['delx'], b(_t['BCy']), b(_t['BCx']),
        _t['delxSqr'], _t['ratioQtr'], _t['ratioSqr'], _t['optArg'], _t['undef'], _lib.hptr(_tf), 1, 0.0,
        ctypes.byref(opt), sp))
    torch.cuda.synchronize()

    for _ in range(a.warmup):
        S.copy_(S0)
        step()
    S.copy_(S0)
    barrier()
    t0 = time.perf_counter()
    ms_sweeps, launches = 0.0, 0
    for _ in range(a.steps):
        s, allf = step()
        ms_sweeps += s['sweep_ms']; launches += s['sweep_launches']
    barrier()
    dt = time.perf_counter() - t0
    if joined:
        tdev = dev if torch.distributed.get_backend() == 'nccl' else torch.device('cpu')
        tt = torch.tensor([dt], dtype=torch.float64, device=tdev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    assert int(allf[0, 2]) == a.sweeps - 1, allf[0]

    if rank == 0:
        total_ps = float(world) * nb * n * a.sweeps * a.steps
        spl = s['sweeps_per_launch']
        avg_ms = ms_sweeps / max(launches, 1)
        alg_bytes = ALG_BYTES['std2d'] * n * nb * spl
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get('std2d_spl%d_um%d' % (spl, s['xuniform_mask']))
            except Exception:
                traffic = None
        out = {
            'metric': 'SOR grid-points*iters/sec (fp64) at %dx%d' % (a.nx, a.ny),
            'value': total_ps / dt, 'unit': 'point-sweeps/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': dt / a.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'invert_Poisson %dx%d lat-lon, land/sea mask, periodic-x, fixed-y '
                                   '(BASELINE configs[1])' % (a.nx, a.ny),
                       'sweeps_per_step': a.sweeps, 'members_per_gpu': nb,
                       'sweeps_per_launch': spl, 'rows_per_tile': s['rows_per_tile'],
                       'xuniform_mask': s['xuniform_mask'], 'masked_tile_pct': s['masked_tile_pct'],
                       'path': {1: 'colour', 2: 'fused'}.get(s['path'], '?'),
                       'parallelism': 'batch-axis shard x%d' % world},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         # measured bytes / launch time: the part of HBM peak really drawn
                         'traffic_GBps': (traffic / (avg_ms * 1e-3) / 1e9) if traffic else None,
                         'traffic_frac': (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         'kernel': 'k_fused2d<FusedStd2D, K=%d, x-uniform mask=%d>' % (spl, s['xuniform_mask']),
                         'avg_launch_ms': avg_ms, 'alg_bytes_per_launch': alg_bytes},
        }
        if world == 1 and not a.no_cpu:
            out['cpu_baseline'] = cpu_baseline(p, a.cpu_seconds)
        print(json.dumps(out))
    if joined:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
