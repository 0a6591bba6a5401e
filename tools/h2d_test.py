import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import util
from xinvert_amd import synthetic
p = synthetic.poisson_latlon(1800, 3600, mask=True)
q = synthetic.member(p, 0)
for rep in range(3):
    t=time.perf_counter()
    S, fl, st = util.run_hip_batched([q], 199, 0.0, shared=(0,1,2))
    dt=time.perf_counter()-t
    print('solve total %.1f ms  h2d %.2f ms  d2h %.2f ms  sweeps %.2f ms' % (dt*1e3, st['h2d_ms'], st['d2h_ms'], st['sweep_ms']))
print('h2d GB/s', 5*51.84e6/ (st['h2d_ms']*1e-3)/1e9, 'd2h GB/s', 51.84e6/(st['d2h_ms']*1e-3)/1e9)
