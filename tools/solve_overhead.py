#!/usr/bin/env python3
"""What a headline solve spends outside its sweep launches: wall clock per solve minus the HIP-event time of the launches.
  python tools/solve_overhead.py [--steps N] [--sweeps S] [--plan 0|1] [--ny --nx]       (XINV_SO selects the library)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--sweeps', type=int, default=500)
    ap.add_argument('--plan', type=int, default=1)
    ap.add_argument('--ny', type=int, default=1800)
    ap.add_argument('--nx', type=int, default=3600)
    ap.add_argument('--members', type=int, default=1)
    a = ap.parse_args()
    import torch
    from xinvert_amd import synthetic
    from xinvert_amd.resident import ResidentProblem
    p = synthetic.poisson_latlon(a.ny, a.nx, mask=True, members=a.members)
    rp = ResidentProblem(p, plan=bool(a.plan))
    for _ in range(5):
        rp.reset(); rp.solve(a.sweeps - 1, 0.0, timing=1)
    rp.reset(); torch.cuda.synchronize()
    t0 = time.perf_counter(); ms = 0.0; nl = 0
    for _ in range(a.steps):
        fl, s = rp.solve(a.sweeps - 1, 0.0, timing=1)
        ms += s['sweep_ms']; nl += s['sweep_launches']
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({'so': os.environ.get('XINV_SO', 'shipped'), 'plan': a.plan, 'ms_per_solve': dt / a.steps * 1e3,
                      'launch_us': ms / nl * 1e3, 'launches_per_solve': nl / a.steps,
                      'ms_outside_launches': (dt * 1e3 - ms) / a.steps,
                      'value': a.members * a.ny * a.nx * a.sweeps * a.steps / dt}))


if __name__ == '__main__':
    main()
