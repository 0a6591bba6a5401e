#!/usr/bin/env python3
"""One short line per JSON line of tools/bench_configs.py (the tables of profiles/r05_seam_rates.txt):
  python tools/bench_configs.py poisson:1800x3601 ... | python tools/r05/row.py <label>"""
import json
import sys

for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print(sys.argv[1], d["config"], d["shape"], "%.4g  launch %.1f us  rows %d K %d pipelined %d"
              % (d["point_sweeps_per_s"], d["avg_launch_ms"] * 1e3, d["rows_per_tile"], d["sweeps_per_launch"], d["pipelined"]))
