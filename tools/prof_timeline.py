#!/usr/bin/env python3
"""Timeline of a rocprofv3 kernel trace (rocpd sqlite): every dispatch that is not the dominant sweep kernel, and the
idle gaps of the GPU between dispatches.   python tools/prof_timeline.py <results.db> [min_gap_us]"""
import sqlite3, sys
db = sys.argv[1]; ming = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = c.execute("select name, start, end from kernels order by start").fetchall()
t0 = rows[0][1]
prev_end = None
tot_gap = 0.0; busy = 0.0
last_big = None
for name, s, e in rows:
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    sweep = name.startswith('void k_pipe2d') or name.startswith('void k_fused2d<FusedStd2D, 1')
    if gap > ming or not sweep:
        print('%10.1f us  gap %7.1f  dur %7.1f  %s' % ((s - t0) / 1e3, gap, (e - s) / 1e3, name[:70]))
    if prev_end is not None and gap > 0: tot_gap += gap
    busy += (e - s) / 1e3
    prev_end = max(prev_end or e, e)
print('dispatches %d, busy %.1f us, gaps %.1f us, span %.1f us' % (len(rows), busy, tot_gap, (rows[-1][2] - t0) / 1e3))
