#!/usr/bin/env python3
"""What one short solve on a resident plan costs (the frame of apps.animate_iteration: 2 sweeps on 73 x 144): the C-ABI
call alone, the Python wrapper around it, and a device-to-device snapshot behind it.
  python tools/r05/frame_cost.py [--n 2000] [--loops 2]"""
import argparse, ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser(); ap.add_argument('--n', type=int, default=2000); ap.add_argument('--loops', type=int, default=2)
ap.add_argument('--ny', type=int, default=73); ap.add_argument('--nx', type=int, default=144)
a = ap.parse_args()
import torch
from xinvert_amd import synthetic, _lib
from xinvert_amd.resident import ResidentProblem
p = synthetic.poisson_latlon(a.ny, a.nx, mask=True)
rp = ResidentProblem(p, plan=True)
for _ in range(20):
    rp.solve(a.loops, 1e-12)
st = torch.cuda.current_stream(rp.dev)
h = rp._plan({}, st)
out = {}
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(a.n):
    rp.L.xinv_plan_solve_f64_dev(h, ctypes.c_void_p(rp.S.data_ptr()), _lib.hptr(rp.flags), a.loops, 1e-12, ctypes.c_void_p(st.cuda_stream))
torch.cuda.synchronize(); out['c_abi_us'] = (time.perf_counter() - t) / a.n * 1e6
t = time.perf_counter()
for _ in range(a.n):
    rp.solve(a.loops, 1e-12)
torch.cuda.synchronize(); out['resident_solve_us'] = (time.perf_counter() - t) / a.n * 1e6
fr = torch.empty((64,) + tuple(rp.S.shape), dtype=rp.S.dtype, device=rp.S.device)
t = time.perf_counter()
for i in range(a.n):
    rp.solve(a.loops, 1e-12); fr[i & 63].copy_(rp.S)
torch.cuda.synchronize(); out['with_snapshot_us'] = (time.perf_counter() - t) / a.n * 1e6
fl, s = rp.solve(a.loops, 1e-12, timing=2)
out['stats'] = {k: s[k] for k in ('sweep_launches', 'launch_us_avg', 'sweeps', 'path') if k in s}
print(json.dumps(out))
