#!/bin/bash
# round 6: k_pipe3d tail cut A/B on ONE box: --cus 1000000 = no remainder cut (the round-5 launch shape)
cd "$GRAFT_REPO_ROOT" || exit 1
out=$PWD/gpurun_out/r06_tail
mkdir -p $out
fmt='import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d["shape"][0], "%.4g" % d["point_sweeps_per_s"], "launch %.1f us" % (d["avg_launch_ms"] * 1e3), {k: d.get(k) for k in ("k_chunks", "cut_tiles")})'
for rep in 1 2; do for m in 13 15 17 9; do for cus in 0 -1; do echo -n "cus=$cus "; python tools/bench_configs.py c5 --members $m --reps 3 --sweeps 100 --cus $cus 2>/dev/null | grep '^{' | python -c "$fmt"; done; done; done | tee $out/c5_tail_ab.txt
