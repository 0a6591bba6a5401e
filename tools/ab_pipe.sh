#!/bin/bash
# A/B of the wave-pipelined kernel: library variants x row-block counts (headline config, timing only)
B="python bench.py --steps 10 --warmup 2 --no-cpu --no-parity --no-hbm"
P='import sys,json; d=json.loads(sys.stdin.read()); print("%.1f G  launch %.1f us  K=%d RY=%d skip %d%%" % (d["value"]/1e9, d["roofline"]["avg_launch_ms"]*1e3, d["config"]["sweeps_per_launch"], d["config"]["rows_per_tile"], d["config"]["masked_tile_pct"]))'
for tag in $1; do
  so=$PWD/build/libxinv_$tag.so; [ "$tag" = cur ] && so=$PWD/xinvert_amd/libxinv_hip.so
  for rows in $2; do
    r=$(XINV_SO=$so timeout 300 $B --rows $rows 2>&1 | tail -1 | python -c "$P")
    echo "$tag rows=$rows: $r"
  done
done
