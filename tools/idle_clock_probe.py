#!/usr/bin/env python3
"""How long an idle GPU has to be idle before the next solve pays for it, and whether trivial work keeps the clocks up:
C4 x 8 resident (7.6 ms warm) after gaps of 0 .. 50 ms spent (a) idle, (b) with a one-block torch kernel running back to back, (c) a streaming kernel over 256 MB, (d) fp64 matrix products.
  python tools/idle_clock_probe.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xinvert_amd import synthetic
from xinvert_amd.resident import ResidentProblem

p = synthetic.gill_matsuno(720, 1440, 8)
rp = ResidentProblem(p)
tiny = torch.zeros(64, device='cuda')
big = torch.zeros(32 << 20, device='cuda', dtype=torch.float64)          # 256 MB: a streaming kernel over it keeps every CU and the HBM busy
ma = torch.randn(2048, 2048, device='cuda', dtype=torch.float64); mb = torch.randn(2048, 2048, device='cuda', dtype=torch.float64)
side = torch.cuda.Stream()


def solve_ms():
    torch.cuda.synchronize(); t = time.perf_counter(); rp.solve(499, 0.0); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3


def gap(ms, busy):
    t0 = time.perf_counter()
    if not busy:
        while (time.perf_counter() - t0) * 1e3 < ms: pass
        return
    with torch.cuda.stream(side):
        while (time.perf_counter() - t0) * 1e3 < ms:
            if busy == 'tiny':
                for _ in range(8): tiny.add_(1.0)
            elif busy == 'stream':
                big.mul_(1.0)
            else:
                torch.mm(ma, mb)
            side.synchronize()


for _ in range(5): rp.reset(); solve_ms()
for busy in (False, 'tiny', 'stream', 'matmul'):
    for g in (0, 1, 4, 16, 50):
        ts = []
        for rep in range(5):
            for _ in range(6): rp.solve(499, 0.0)          # warm
            torch.cuda.synchronize()
            gap(g, busy)
            ts.append(solve_ms())
        ts.sort()
        print(json.dumps({'gap_ms': g, 'during the gap': busy or 'idle', 'solve_ms median / min': [round(ts[2], 3), round(ts[0], 3)]}), flush=True)
