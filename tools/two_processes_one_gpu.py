#!/usr/bin/env python3
"""Two processes on ONE GPU solving two C5 volumes each at the same time: are the fields those of a process alone?"""
import os, sys, subprocess, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == 'worker':
    lo, hi, cus, sweeps, lanes = (int(v) for v in sys.argv[2:7])
    import numpy as np, torch
    import bench
    from xinvert_amd.resident import ResidentProblem
    q = bench.c5_members(lo, hi)
    rp = ResidentProblem(q)
    open('/tmp/conc_ready_%d' % lo, 'w').write('1')
    while not (os.path.exists('/tmp/conc_go')):
        time.sleep(0.01)
    for rep in range(3):
        rp.reset(); rp.solve(sweeps - 1, 0.0, cu_count=cus, lanes=lanes); torch.cuda.synchronize()
        print(lo, hi, 'cus', cus, 'lanes', lanes, 'rep', rep, rp.S.view(torch.int64).reshape(rp.nb, -1).sum(dim=1).cpu().numpy().tolist(), flush=True)
    sys.exit(0)
for cus, lanes in ((0, 0), (-1, 0), (0, 1), (-1, 1)):
    for conc in (1, 0):
        for f in ('/tmp/conc_go', '/tmp/conc_ready_0', '/tmp/conc_ready_2'):
            if os.path.exists(f): os.remove(f)
        print('---- cu_count', cus, 'lanes', lanes, 'concurrent' if conc else 'one after the other', flush=True)
        if conc:
            ps = [subprocess.Popen([sys.executable, __file__, 'worker', str(lo), str(lo + 2), str(cus), '11', str(lanes)]) for lo in (0, 2)]
            while not (os.path.exists('/tmp/conc_ready_0') and os.path.exists('/tmp/conc_ready_2')):
                time.sleep(0.05)
            open('/tmp/conc_go', 'w').write('1')
            for p in ps: p.wait()
        else:
            open('/tmp/conc_go', 'w').write('1')
            for lo in (0, 2):
                subprocess.call([sys.executable, __file__, 'worker', str(lo), str(lo + 2), str(cus), '11', str(lanes)])
