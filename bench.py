#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X SOR inversion engine.

Metric (BASELINE.json): SOR grid-points x iterations per second, fp64, on the 3600 x 1800 global
lat-lon Poisson problem with a land/sea mask (configs[1]).  One *step* is one complete hot-path
pass: `xinv_standard_2d_f64_dev` over one batch of synthetic input already resident in HBM,
running a fixed number of sweeps (tolerance = 0, mxLoop = sweeps - 1), norm + stopping rule
evaluated on the device after every sweep exactly as in production.  Every point of the grid is
counted, masked (land) points included -- `value_active` counts only tiles the kernel ran.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c4|c5] [--sweeps S] ...

--config c2 (default): N > 1 = every rank solves its own member(s) (weak scaling).
--config c4 / c5: the REAL batch of BASELINE configs[3] / [4] (64 Gill-Matsuno members /
120 omega volumes) split in contiguous blocks over the ranks (strong scaling).
N > 1 is launched by torch.distributed.run, one rank per GPU; no data-path collective, the
per-slice flags are all-gathered over RCCL after each step.  Rank 0 prints ONE JSON line.

Extra objects in the JSON line (N = 1, c2):
  roofline       the resource that bounds the dominant kernel: the fp64 vector ALU.  achieved =
                 useful point updates x 16 fp64 operations / mean launch duration (HIP events on the
                 solve's stream over the timed region); peak = fp64 VALU issue rate WITHOUT FMA
                 (the bit-exactness contract forbids contraction); `executed_over_useful` is the
                 recomputed-halo factor of the tiling actually used.  `alg_equiv_GBps` is SURVEY
                 8(d)'s 48 B/point figure over the same time -- a comparable number, NOT a fraction
                 of anything (the kernel elides B, reads A and C as per-row scalars and fuses K
                 sweeps per pass); `traffic` is the PMC-measured bytes per launch of the same kernel
                 variant, read from profiles/traffic.json (static: counters cannot be read in-process).
  roofline_hbm   the HBM-bound variant north_star names, measured in the same run: one sweep per
                 pass, every coefficient array streamed in full, every tile run.  achieved =
                 48 B x points / launch duration; frac <= 1 by construction.
  parity         after the timed loop the same solve is repeated from the initial state and compared
                 BIT FOR BIT with the CPU oracle's coloured ordering run for the same sweeps.
  cpu_baseline   the oracle's lexicographic sweep (the reference's execution model) built with
                 -march=native on this box, 1 core and all cores.
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
# fp64 operations of one point update as the kernels execute it (the relaxation factor and
# F*delxSqr are hoisted per row / per launch): std2d 4 sub + 4 mul + 2 sub + mul + add + sub + mul + add = 16;
# gen2d (hoisted) 25; std3d (hoisted, f*delxSqr per point) 21
UPD_FLOPS = {'std2d': 16, 'gen2d': 25, 'std3d': 21}
HBM_PEAK_GBS = 8000.0                                       # MI355X_MICROARCH.md: 8.0 TB/s spec
# fp64 vector ALU: 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz = 39.3 T operations/s (78.6 TFLOP/s
# datasheet figure counts an FMA as two; tools/fp64_peak.hip measures both on the box)
FP64_VALU_PEAK_TFLOPS = 39.3
FP64_VALU_MEASURED_TFLOPS = 34.0                             # profiles/r02_fp64_peak.txt (mul+add chains, no FMA)
FP64_FMA_SPEC_TFLOPS = 78.6


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default='c2', choices=['c2', 'c4', 'c5'])
    ap.add_argument('--sweeps', type=int, default=0, help='SOR sweeps per step (0: SURVEY.md 8(d): 500 for c2/c4, 200 for c5)')
    ap.add_argument('--spl', type=int, default=0, help='sweeps fused per launch (0 = engine default)')
    ap.add_argument('--rows', type=int, default=0, help='rows per tile (0 = engine default)')
    ap.add_argument('--members', type=int, default=0, help='c2: batch members per GPU (default 1); c4/c5: total batch (default 64 / 120)')
    ap.add_argument('--ny', type=int, default=1800)
    ap.add_argument('--nx', type=int, default=3600)
    ap.add_argument('--no-cpu', action='store_true', help='skip the cpu_baseline leg')
    ap.add_argument('--no-parity', action='store_true', help='skip the post-run oracle parity check')
    ap.add_argument('--no-hbm', action='store_true', help='skip the HBM-bound variant')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--parity-sweeps', type=int, default=0, help='sweeps of the parity check (0 = the timed count)')
    return ap.parse_args()


def cpu_baseline(q, budget_s):
    """Time the oracle's lexicographic sweep (the reference's execution model: one slice, one
    core) on member 0 of the same workload for about `budget_s` seconds, with the -march=native
    build of the oracle made on THIS box; falls back to the travelling build."""
    import oracle as orc
    native = orc.use_native()
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
    return out


def oracle_parity(q, S_hip, flags_hip, sweeps):
    """Bitwise comparison of the HIP result with the oracle's coloured ordering (same input, same
    sweep count).  The checker, run after the timed region."""
    import oracle as orc
    orc.use_portable()                   # parity is defined against the -ffp-contract=off portable build
    c = [np.ascontiguousarray(a, dtype=np.float64) for a in q['coefs']]
    S = np.array(q['S0'], dtype=np.float64, copy=True)
    fl = np.array([0., 1., 0.])
    t = time.perf_counter()
    orc.standard_2d(S, *c, q['yc'], q['xc'], q['dely'], q['delx'], q['BCy'], q['BCx'], q['delxSqr'],
                    q['ratioQtr'], q['ratioSqr'], q['optArg'], q['undef'], fl, sweeps - 1, 0.0, orc.COLOUR_2)
    dt = time.perf_counter() - t
    same = bool(np.array_equal(S, S_hip))
    nbad = int((S != S_hip).sum())
    return {'bitwise': same, 'sweeps': int(sweeps), 'mismatching_points': nbad,
            'loop_equal': bool(fl[2] == flags_hip[2]),
            'flag1_abs_diff': float(abs(fl[1] - flags_hip[1])),
            'against': 'oracle/xinv_oracle.c coloured (red-black) ordering, %.1f s on 1 core' % dt}


def tile_model(s, ny, nx):
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


def build_problem(a, rank, world):
    """-> (problem dict restricted to this rank's members, total members over all ranks, scaling)."""
    from xinvert_amd import synthetic
    from xinvert_amd import dist as xdist
    if a.config == 'c2':
        nb = a.members or 1
        p = synthetic.poisson_latlon(a.ny, a.nx, mask=True, seed=synthetic.SEED + rank, members=nb)
        return p, nb * world, 'weak', 'invert_Poisson %dx%d lat-lon, land/sea mask, periodic-x, fixed-y (BASELINE configs[1])' % (a.nx, a.ny)
    total = a.members or (64 if a.config == 'c4' else 120)
    lo, hi = xdist.shard_range(total, rank, world)
    if a.config == 'c4':
        # every rank draws the same member list (same seed) and keeps its block
        p = synthetic.gill_matsuno(720, 1440, total)
        p['S0'] = p['S0'][lo:hi]
        p['coefs'] = [c if k in p['shared'] else c[lo:hi] for k, c in enumerate(p['coefs'])]
        return p, total, 'strong', 'invert_GillMatsuno 1440x720, %d forcing members (BASELINE configs[3])' % total
    # c5: generated block by block (a 120-step forcing is 12 GB per array on the host)
    parts = []
    for m0 in range(lo, hi, 8):
        parts.append(synthetic.omega_latlon(50, 360, 720, steps=min(8, hi - m0), seed=synthetic.SEED + m0))
    p = dict(parts[0])
    p['S0'] = np.concatenate([q['S0'] for q in parts])
    p['coefs'] = [c if k in p['shared'] else np.concatenate([q['coefs'][k] for q in parts])
                  for k, c in enumerate(p['coefs'])]
    return p, total, 'strong', 'invert_omega 720x360x50, %d time steps (BASELINE configs[4])' % total


def main():
    a = parse()
    import torch
    from xinvert_amd import _lib, synthetic
    from xinvert_amd import dist as xdist
    from xinvert_amd.resident import ResidentProblem

    rank, local, world = xdist.init_process_group()
    joined = torch.distributed.is_available() and torch.distributed.is_initialized()
    if world != a.gpus and world > 1:
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
    rp.reset()
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
    assert (allf[:, 2] == sweeps - 1).all() and not allf[:, 0].any(), allf[:4]

    if rank == 0:
        total_ps = float(total_members) * n * sweeps * a.steps
        spl = s['sweeps_per_launch']
        avg_ms = ms_sweeps / max(launches, 1)
        # the timed launches: full K-sweep passes plus one shorter tail pass per step when K does
        # not divide the sweep count; per-launch figures below use the mean over all of them
        sweeps_per_launch_mean = float(sweeps) * a.steps / max(launches, 1)
        upd_per_launch = float(nb) * n * sweeps_per_launch_mean
        out = {
            'metric': 'SOR grid-points*iters/sec (fp64) at %dx%d%s, masked points counted'
                      % ((a.nx, a.ny, '') if a.config == 'c2' else
                         ((1440, 720, ' x %d members' % total_members) if a.config == 'c4' else
                          (720, 360, 'x50 x %d steps' % total_members))),
            'value': total_ps / dt, 'unit': 'point-sweeps/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': dt / a.steps * 1e3, 'higher_is_better': True, 'scaling': scaling,
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': wl_name, 'sweeps_per_step': sweeps,
                       'members_total': total_members, 'members_this_gpu': nb,
                       'sweeps_per_launch': spl, 'rows_per_tile': s['rows_per_tile'],
                       'xuniform_mask': s['xuniform_mask'], 'masked_tile_pct': s['masked_tile_pct'],
                       'path': {1: 'colour', 2: 'fused'}.get(s['path'], '?'),
                       'parallelism': 'batch-axis shard x%d' % world},
        }
        if s['masked_tile_pct']:
            out['value_active'] = out['value'] * (1.0 - s['masked_tile_pct'] / 100.0)
        flops = UPD_FLOPS[kind] * upd_per_launch
        achieved_tf = flops / (avg_ms * 1e-3) / 1e12
        roof = {'bound': 'valu_fp64', 'achieved': achieved_tf, 'peak': FP64_VALU_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': achieved_tf / FP64_VALU_PEAK_TFLOPS,
                'peak_note': 'fp64 vector issue rate without FMA (256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz); '
                             'contraction is off by the bit-exactness contract; datasheet FMA peak %.1f'
                             % FP64_FMA_SPEC_TFLOPS,
                'frac_of_fma_spec': achieved_tf / FP64_FMA_SPEC_TFLOPS,
                'frac_of_measured_peak': achieved_tf / FP64_VALU_MEASURED_TFLOPS,
                'useful_flops_per_point_update': UPD_FLOPS[kind],
                'kernel': ('k_pipe2d<NP=%d> (four sweeps per pass, one per wavefront; x-uniform mask=%d)'
                           % (s['pipelined'], s['xuniform_mask'])) if s.get('pipelined')
                          else ('k_fused2d<FusedStd2D, K=%d, x-uniform mask=%d>' % (spl, s['xuniform_mask'])) if kind == 'std2d'
                          else ('k_fused2d<FusedGen2D, K=%d, x-uniform mask=%d>' % (spl, s['xuniform_mask'])) if kind == 'gen2d'
                          else 'k_fused3d',
                'avg_launch_ms': avg_ms, 'launches': int(launches),
                'alg_bytes_per_launch': ALG_BYTES[kind] * upd_per_launch,
                'alg_equiv_GBps': ALG_BYTES[kind] * upd_per_launch / (avg_ms * 1e-3) / 1e9}
        if kind != 'std3d':
            em = tile_model(s, p['yc'], p['xc'])
            if em:
                roof['executed_over_useful'] = em
                roof['frac_executed'] = roof['frac'] * em
        if kind == 'std3d':
            # the 3-D kernels are bound by what the fabric delivers, not by the VALU: S read + S write +
            # the forcing + every coefficient array that is not a per-row scalar, 8 B each
            nstream = 3 + (3 - bin(s['xuniform_mask'] & 7).count('1'))
            vb = 8.0 * nstream * upd_per_launch
            roof.update({'bound': 'hbm', 'achieved': vb / (avg_ms * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': vb / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         'bytes_note': 'bytes the kernel variant must move per point-sweep: %d streams x 8 B '
                                       '(x-uniform coefficient arrays are per-row scalars); SURVEY 8(d) algorithmic '
                                       'figure kept as alg_equiv_GBps' % nstream,
                         'valu_TFLOPs': achieved_tf, 'valu_frac': achieved_tf / FP64_VALU_PEAK_TFLOPS})
        traffic = None
        tfile = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tfile) and a.config == 'c2' and (a.ny, a.nx) == (1800, 3600) and nb == 1:
            try:
                traffic = json.load(open(tfile)).get(('std2d_pipe_um%d' % s['xuniform_mask']) if s.get('pipelined')
                                                     else 'std2d_spl%d_um%d' % (spl, s['xuniform_mask']))
            except Exception:
                traffic = None
        roof['traffic'] = traffic
        roof['traffic_source'] = 'static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel variant (profiles/traffic.json), not read in this run'
        if traffic:
            roof['traffic_GBps'] = traffic / (avg_ms * 1e-3) / 1e9
            roof['traffic_frac_of_hbm_peak'] = roof['traffic_GBps'] / HBM_PEAK_GBS
        out['roofline'] = roof

        if a.config == 'c2' and world == 1 and not a.no_hbm:
            # ---- the HBM-bound variant: K = 1, every array streamed, every tile run -----------
            hb = ResidentProblem(p, device=local, null_zero_B=False)
            hopts = dict(sweeps_per_launch=1, timing=1, no_xuniform=1, no_tile_skip=1)
            hsw = 100
            hb.solve(hsw - 1, 0.0, **hopts)
            hb.reset()
            torch.cuda.synchronize()
            th = time.perf_counter()
            hms, hl = 0.0, 0
            for _ in range(5):
                fl_h, sh = hb.solve(hsw - 1, 0.0, **hopts)
                hms += sh['sweep_ms']; hl += sh['sweep_launches']
            torch.cuda.synchronize()
            th = time.perf_counter() - th
            h_avg = hms / max(hl, 1)
            h_ach = ALG_BYTES[kind] * float(nb) * n / (h_avg * 1e-3) / 1e9
            htraffic = None
            try:
                htraffic = json.load(open(tfile)).get('std2d_spl1_um0_all')
            except Exception:
                pass
            out['roofline_hbm'] = {
                'bound': 'hbm', 'achieved': h_ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': h_ach / HBM_PEAK_GBS,
                'variant': 'k_fused2d<FusedStd2D, K=1, x-uniform mask=0>: one sweep per pass, S A C F streamed in full '
                           '(B is identically zero: detected, not re-read per sweep; still counted in the 48 B), no tile skipping',
                'value': float(nb) * n * hsw * 5 / th, 'unit_value': 'point-sweeps/s',
                'avg_launch_ms': h_avg, 'launches': int(hl), 'alg_bytes_per_launch': ALG_BYTES[kind] * float(nb) * n,
                'traffic': htraffic, 'traffic_source': roof['traffic_source'],
                'traffic_GBps': (htraffic / (h_avg * 1e-3) / 1e9) if htraffic else None,
                'traffic_frac_of_hbm_peak': (htraffic / (h_avg * 1e-3) / 1e9 / HBM_PEAK_GBS) if htraffic else None,
                'note': 'S (two buffers) + A + C + F = 259 MB fit the 256 MiB Infinity Cache almost entirely: the measured '
                        'bytes are fabric traffic, the HBM itself is touched less',
                'sweeps_per_launch': sh['sweeps_per_launch'], 'xuniform_mask': sh['xuniform_mask'],
                'masked_tile_pct': sh['masked_tile_pct']}
            del hb

        if a.config == 'c2' and world == 1 and not a.no_parity:
            psw = a.parity_sweeps or sweeps
            rp.reset()
            fl_p, sp_ = rp.solve(psw - 1, 0.0, **opts)
            same_cfg = all(sp_[k] == s[k] for k in ('path', 'sweeps_per_launch', 'rows_per_tile',
                                                    'xuniform_mask', 'masked_tile_pct'))
            q = synthetic.member(p, 0)
            par = oracle_parity(q, rp.result()[0], fl_p[0], psw)
            par['same_kernel_config_as_timed'] = bool(same_cfg)
            out['parity'] = par
        if a.config == 'c2' and world == 1 and not a.no_cpu:
            out['cpu_baseline'] = cpu_baseline(synthetic.member(p, 0), a.cpu_seconds)
        print(json.dumps(out))
    if joined:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
