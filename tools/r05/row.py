import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[1], d["config"], d["shape"], "%.4g  launch %.1f us  rows %d K %d pipelined %d" % (d["point_sweeps_per_s"], d["avg_launch_ms"]*1e3, d["rows_per_tile"], d["sweeps_per_launch"], d["pipelined"]))
