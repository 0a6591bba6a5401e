#!/usr/bin/env python3
"""Why are the first solves after ResidentProblem.reset() slower (5.3, 4.8, 4.6, ... 4.25 ms at 3600 x 1800)?  The time of a 500-sweep
solve against what S holds when it starts: the first guess (zeros), a field that has been swept for a while, the same after an idle gap.
  python tools/step_time_vs_state.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xinvert_amd import synthetic
from xinvert_amd.resident import ResidentProblem

p = synthetic.poisson_latlon(1800, 3600)
rp = ResidentProblem(p)


def steps(k):
    out = []
    for _ in range(k):
        torch.cuda.synchronize(); t = time.perf_counter(); rp.solve(499, 0.0); torch.cuda.synchronize(); out.append(round((time.perf_counter() - t) * 1e3, 3))
    return out


rp.reset(); steps(3)
rp.reset(); print(json.dumps({'after reset (S = first guess)': steps(12)}))
print(json.dumps({'continuing (no reset)': steps(6)}))
swept = rp.S.clone()
time.sleep(0.5)
print(json.dumps({'continuing after 0.5 s of idle': steps(6)}))
rp.reset(); torch.cuda.synchronize(); rp.S.copy_(swept); torch.cuda.synchronize()
print(json.dumps({'after reset, then S overwritten with the swept field': steps(6)}))
rp.reset(); print(json.dumps({'after reset again': steps(8)}))
# the state after n sweeps: how long does the NEXT solve take
for n in (500, 2000, 8000):
    rp.reset(); rp.solve(n - 1, 0.0); print(json.dumps({'after %d sweeps from the first guess' % n: steps(3)}))
