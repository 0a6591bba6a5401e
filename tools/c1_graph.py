import sys, time, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xinvert_amd import synthetic
from xinvert_amd.resident import ResidentProblem
p = synthetic.poisson_latlon(180, 360)
rp = ResidentProblem(p)
n = 180 * 360
for opt in ({}, {'graph': 1}, {'graph': 1, 'check_every': 16}, {'graph': 1, 'check_every': 62}, {'graph': 1, 'check_every': 124}, {'norm_lag': -1}, {'graph': -1}):
    best = 1e9
    for rep in range(6):
        rp.reset(); torch.cuda.synchronize()
        t = time.perf_counter(); fl, st = rp.solve(499, 0.0, **opt); torch.cuda.synchronize(); dt = time.perf_counter() - t
        if rep: best = min(best, dt)
    print(json.dumps({'opt': opt, 'ms': round(best * 1e3, 4), 'value': n * 500 / best, 'launches': st['sweep_launches'], 'flags': fl[0].tolist()}))
