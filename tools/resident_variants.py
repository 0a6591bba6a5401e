#!/usr/bin/env python3
"""A device-resident batch under the launch structures the host-pointer pipeline uses (lanes, lagged norm, concurrent solves of
parts of the batch from several host threads): which of them costs what.   python tools/resident_variants.py [members] [sweeps]"""
import json, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xinvert_amd import synthetic
from xinvert_amd.resident import ResidentProblem

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
p = synthetic.gill_matsuno(720, 1440, nb)
n = 720 * 1440


def timed(fn, reps=4):
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return best * 1e3


rp = ResidentProblem(p)
for opt in ({}, {'lanes': 1}, {'lanes': 2}, {'lanes': 4}, {'norm_lag': -1}, {'norm_lag': -1, 'lanes': 1}):
    def run():
        rp.reset(); torch.cuda.synchronize()
    run()
    rp.solve(sweeps - 1, 0.0, **opt)
    ms = []
    for _ in range(4):
        rp.reset(); torch.cuda.synchronize()
        t = time.perf_counter(); fl, st = rp.solve(sweeps - 1, 0.0, **opt); torch.cuda.synchronize(); ms.append((time.perf_counter() - t) * 1e3)
    print(json.dumps({'whole batch': nb, 'options': opt, 'ms': round(min(ms), 3), 'lanes': st['lanes'], 'launches': st['sweep_launches']}), flush=True)
del rp
# the batch as `parts` resident problems solved concurrently, one host thread and one stream each (the chunk scheme's structure)
for parts in (2, 4, 8):
    if nb % parts: continue
    per = nb // parts
    rps = [ResidentProblem(p, members=(k * per, (k + 1) * per)) for k in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    for lanes in (0, 1):
        def one(k):
            rps[k].solve(sweeps - 1, 0.0, stream=streams[k], lanes=lanes)
        for k in range(parts): one(k)
        torch.cuda.synchronize()
        ms = []
        for _ in range(4):
            for r in rps: r.reset()
            torch.cuda.synchronize()
            t = time.perf_counter()
            th = [threading.Thread(target=one, args=(k,)) for k in range(parts)]
            for x in th: x.start()
            for x in th: x.join()
            torch.cuda.synchronize(); ms.append((time.perf_counter() - t) * 1e3)
        print(json.dumps({'parts': parts, 'members each': per, 'threads': parts, 'lanes option': lanes, 'ms': round(min(ms), 3)}), flush=True)
    # ... and issued by ONE thread, one after the other (each solve returns when its chain has been queued and finished?)
    del rps
