#!/usr/bin/env python3
"""The small slice (C1, 360 x 180) after an idle gap / after a burst of fp64 matrix products / back to back: how much of its
latency-bound launch chain is the clock state (profiles/r06_idle_clocks.txt (5)).   python tools/c1_clock_probe.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xinvert_amd import synthetic
from xinvert_amd.resident import ResidentProblem
p = synthetic.poisson_latlon(180, 360)
rp = ResidentProblem(p)
ma = torch.randn(2048, 2048, device='cuda', dtype=torch.float64); mb = torch.randn(2048, 2048, device='cuda', dtype=torch.float64)
def solve_ms():
    t = time.perf_counter(); rp.solve(499, 0.0); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
def burst(ms):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms: torch.mm(ma, mb); torch.cuda.synchronize()
for _ in range(5): rp.reset(); solve_ms()
for what in ('idle 30 ms', 'fp64 matmul 30 ms', 'back to back'):
    ts = []
    for rep in range(7):
        rp.reset(); torch.cuda.synchronize()
        if what.startswith('idle'): time.sleep(0.03)
        elif what.startswith('fp64'): burst(30)
        else:
            for _ in range(20): rp.solve(499, 0.0)
            torch.cuda.synchronize()
        ts.append(solve_ms())
    ts.sort(); print(json.dumps({'C1 500 sweeps after': what, 'ms median / min': [round(ts[3], 4), round(ts[0], 4)]}))
# ten solves in a row after idle: does a latency-bound chain of small launches bring the clocks up by itself?
rp.reset(); torch.cuda.synchronize(); time.sleep(0.05)
print(json.dumps({'ten solves in a row after 50 ms idle': [round(solve_ms(), 4) for _ in range(10)]}))
burst(40)
print(json.dumps({'ten solves in a row after 40 ms of matmul': [round(solve_ms(), 4) for _ in range(10)]}))
