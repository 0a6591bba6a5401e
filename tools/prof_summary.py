#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2, rocpd sqlite output) runs into small text/JSON files.

  python tools/prof_summary.py kernels  <results.db> [out.txt]     per-kernel calls / avg / min / max us
  python tools/prof_summary.py counters <results.db> [out.txt]     per-kernel mean of each PMC counter
  python tools/prof_summary.py config   <fetch.db> <write.db> <kernel-substring> <config name> <bench.py kernel prefix>
                                        <members> <point-sweeps per launch> <traffic.json> [source text]
        the same per point-sweep, stored under traffic.json['configs'][name] (bench.py's per-configuration lines)
  python tools/prof_summary.py traffic  <fetch.db> <write.db> <kernel-substring> <key> [traffic.json [members lanes]]
        HBM-side bytes per launch of one kernel = 2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes); with members / lanes the
        entry also records bytes per PASS (a pass over `members` in `lanes` launch chains = lanes kernel launches).
        The factor 2 on FETCH_SIZE is the gfx950 correction of MI355X_MICROARCH.md (HBM section):
        this rocprofv3 tallies 128-B read requests at 64 B.  It is re-checked in every profile by
        k_strip_active / k_any_nonzero, which stream exactly one array (known byte count).
"""
import json
import os
import sqlite3
import sys


def kernels(db):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), avg(duration), min(duration), max(duration), sum(duration), "
                     "max(vgpr_count), max(sgpr_count), max(grid_x), max(workgroup_x) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[5] for r in rows) or 1
    out = ['%-64s %7s %10s %10s %10s %7s %5s %5s %9s' % ('kernel', 'calls', 'avg_us', 'min_us', 'max_us', 'pct', 'vgpr', 'sgpr', 'grid_x')]
    for r in rows:
        out.append('%-64s %7d %10.2f %10.2f %10.2f %6.2f%% %5d %5d %9d' % (
            r[0][:64], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, 100.0 * r[5] / tot, r[6], r[7], r[8]))
    return '\n'.join(out)


def counters(db):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) "
                     "from counters_collection group by kernel_name, counter_name order by kernel_name").fetchall()
    out = ['%-64s %-18s %7s %16s %16s %16s' % ('kernel', 'counter', 'n', 'mean', 'min', 'max')]
    for r in rows:
        out.append('%-64s %-18s %7d %16.3f %16.3f %16.3f' % (r[0][:64], r[1], r[2], r[3], r[4], r[5]))
    return '\n'.join(out)


def mean_counter(db, name, kernel_sub):
    c = sqlite3.connect(db)
    r = c.execute("select avg(value), count(*) from counters_collection where counter_name=? and kernel_name like ?",
                  (name, '%' + kernel_sub + '%')).fetchone()
    return r[0], r[1]


def src_sha():
    """hash of the library's sources as they are in this tree (xinvert_amd/build.py: source_hash) -- every entry written
    into traffic.json carries it, and bench.py reports the entry only while it matches the tree it runs from"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from xinvert_amd import build as xbuild
    return xbuild.source_hash()


def main():
    cmd = sys.argv[1]
    if cmd in ('kernels', 'counters'):
        txt = kernels(sys.argv[2]) if cmd == 'kernels' else counters(sys.argv[2])
        if len(sys.argv) > 3:
            open(sys.argv[3], 'w').write(txt + '\n')
        print(txt)
    elif cmd == 'traffic':
        fdb, wdb, ksub, key = sys.argv[2:6]
        f, nf = mean_counter(fdb, 'FETCH_SIZE', ksub)
        w, nw = mean_counter(wdb, 'WRITE_SIZE', ksub)
        # calibration: a kernel that streams exactly one yc*xc array once -- k_strip_active reads
        # the forcing (every solve with masked-tile skipping), k_any_nonzero a coefficient array
        cal, _ = mean_counter(fdb, 'FETCH_SIZE', 'k_strip_active')
        calname = 'k_strip_active'
        if not cal:
            cal, _ = mean_counter(fdb, 'FETCH_SIZE', 'k_any_nonzero')
            calname = 'k_any_nonzero'
        traffic = (2.0 * f + w) * 1024.0
        print('%s: FETCH_SIZE %.1f KiB (n=%d) x2, WRITE_SIZE %.1f KiB (n=%d) -> %.4e B per launch'
              % (ksub, f, nf, w, nw, traffic))
        if cal:
            print('calibration: %s FETCH_SIZE %.1f KiB (streams one yc*xc array)' % (calname, cal))
        if len(sys.argv) > 6:
            path = sys.argv[6]
            d = json.load(open(path)) if os.path.exists(path) else {}
            d[key] = traffic
            d[key + '_detail'] = {'FETCH_SIZE_KiB': f, 'WRITE_SIZE_KiB': w, 'fetch_correction': 2.0,
                                  'calibration_kernel': calname, 'calibration_FETCH_KiB': cal,
                                  'bytes_per_launch': traffic, 'launches_profiled': nf, 'src_sha': src_sha()}
            if len(sys.argv) > 8:                          # members of the whole pass, lanes it ran in (a launch covers members / lanes)
                d[key + '_detail'].update({'members': int(sys.argv[7]), 'lanes': int(sys.argv[8]),
                                           'bytes_per_pass': traffic * int(sys.argv[8])})
            json.dump(d, open(path, 'w'), indent=1, sort_keys=True)
    elif cmd == 'config':
        fdb, wdb, ksub, name, prefix, members, psl, path = sys.argv[2:10]
        src = sys.argv[10] if len(sys.argv) > 10 else 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes'
        f, nf = mean_counter(fdb, 'FETCH_SIZE', ksub)
        w, nw = mean_counter(wdb, 'WRITE_SIZE', ksub)
        traffic = (2.0 * f + w) * 1024.0
        d = json.load(open(path)) if os.path.exists(path) else {}
        d.setdefault('configs', {})[name] = {'kernel_prefix': prefix, 'members': int(members),
                                             'bytes_per_point_sweep': traffic / float(psl), 'bytes_per_launch': traffic,
                                             'FETCH_SIZE_KiB': f, 'WRITE_SIZE_KiB': w, 'launches_profiled': nf, 'source': src,
                                             'src_sha': src_sha()}
        json.dump(d, open(path, 'w'), indent=1, sort_keys=True)
        print('%s: %s  %.4e B per launch = %.2f B per point-sweep (n=%d)' % (name, ksub, traffic, traffic / float(psl), nf))
    else:
        raise SystemExit(__doc__)


if __name__ == '__main__':
    main()
