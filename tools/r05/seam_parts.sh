#!/bin/bash
# Odd-xc periodic seam: how many pieces the edge strips' row blocks are cut into (xinv_tile_rows), measured on the
# experiments build (XINV_SEAM_PARTS is read only there):
#   XINV_BUILD_TAG=exp XINV_EXTRA_FLAGS=-DXINV_EXPERIMENTS XINV_VARIANT_UNITS=xinv_hip python -m xinvert_amd.build
#   bash tools/r05/seam_parts.sh > gpurun_out/r05k/seam_parts.txt
cd "${GRAFT_REPO_ROOT:-$PWD}" || exit 1
export XINV_SO=$PWD/build/libxinv_exp.so
row() { python tools/bench_configs.py "$@" --reps 3 2>/dev/null | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('  ', d['config'], d['shape'], '%.4g  launch %.1f us  rows %d K %d pipelined %d' % (d['point_sweeps_per_s'], d['avg_launch_ms']*1e3, d['rows_per_tile'], d['sweeps_per_launch'], d['pipelined']))"; }
for parts in 2 3 4 5; do
  echo "XINV_SEAM_PARTS=$parts"
  export XINV_SEAM_PARTS=$parts
  row poisson:1800x3600 poisson:1800x3601 poisson:1801x3601
  row poisson:1800x3600 poisson:1800x3601 --members 8
  row gm:720x1440 gm:720x1441 --members 8
done
