import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_parity as t
import util
cap = {}
orig = t.run_hip_batched
def wrap(ps, nsw, tol, **kw):
    cap['args'] = (ps, nsw, tol, kw)
    return orig(ps, nsw, tol, **kw)
t.run_hip_batched = wrap
chunk, s0, odd = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
try:
    t._medium_fuzz(chunk, s0, odd)
    print('no failure')
except Exception as e:
    print('failed as expected:', str(e)[:150])
ps, nsw, tol, kw = cap['args']
print('kind', ps[0]['kind'], ps[0]['S0'].shape, ps[0]['BCy'], ps[0]['BCx'], 'nsw', nsw, 'tol', tol, kw)
from oracle import COLOUR_AUTO
for m, q in enumerate(ps):
    So, flo = util.run_oracle(q, nsw, tol, COLOUR_AUTO)
    print('member', m, 'oracle flags', flo, 'max|S|', np.nanmax(np.abs(np.where(So == q['undef'], 0, So))))
S, fl, st = orig(ps, nsw, tol, **kw)
print('hip flags', fl, st['path'], st['rows_per_tile'])
for m, q in enumerate(ps):
    So, flo = util.run_oracle(q, nsw, tol, COLOUR_AUTO)
    d = S[m] != So
    print('member', m, 'differ', d.sum(), 'rows', np.unique(np.where(d)[0]), 'ncols', len(np.unique(np.where(d)[1])))
# fixed sweep counts, no tolerance: where does it start?
for k in range(1, 9):
    S, fl, st = orig(ps, k, 0.0, **kw)
    for m, q in enumerate(ps[:1]):
        So, flo = util.run_oracle(q, k, 0.0, COLOUR_AUTO)
        d = S[m] != So
        print('mxLoop', k, 'member', m, 'differ', d.sum(), 'rows', np.unique(np.where(d)[0])[:8], 'hip fl', fl[m], 'orc fl', flo)
# single member
S, fl, st = orig(ps[:1], nsw, tol, **kw)
So, flo = util.run_oracle(ps[0], nsw, tol, COLOUR_AUTO)
d = S[0] != So
print('single member: differ', d.sum(), fl[0], flo, 'lanes', st['lanes'])
S, fl, st = orig(ps, nsw, tol, path=1, **{k: v for k, v in kw.items() if k != 'force_tile_skip'})
d = S[0] != util.run_oracle(ps[0], nsw, tol, COLOUR_AUTO)[0]
print('colour path: differ', d.sum())
