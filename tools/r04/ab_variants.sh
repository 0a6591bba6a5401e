#!/bin/bash
# A/B harness of round 4 (what the one-off session drivers run1..run17 did, kept as ONE script):
#   bash tools/r04/ab_variants.sh <bench_configs args ...> -- [build/libxinv_<tag>.so ...]
# for the shipped library and every variant library named (xinvert_amd/build.py: XINV_BUILD_TAG / XINV_VARIANT_UNITS /
# XINV_EXTRA_FLAGS make them): throughput (best of 3), the two-sweep 3-D parity tests, and the HBM-side traffic of the
# dominant kernel (separate FETCH_SIZE / WRITE_SIZE passes, 2 x FETCH + WRITE: MI355X_MICROARCH.md).
# Example (profiles/r04_pipe3d_variants.txt):  bash tools/r04/ab_variants.sh c5 --members 15 -- build/libxinv_p3frm.so
cd "${GRAFT_REPO_ROOT:-$PWD}" || exit 1
export TMPDIR=/tmp
R=$PWD
args=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
db() { find "$1" -name '*.db' | head -1; }
for so in "" "$@"; do
  tag=$(basename "${so:-shipped}" .so)
  if [ -n "$so" ]; then export XINV_SO=$R/$so; else unset XINV_SO; fi
  python tools/bench_configs.py "${args[@]}" --reps 3 2>/dev/null | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$tag', d['config'], d['shape'], '%.4g  launch %.1f us' % (d['point_sweeps_per_s'], d['avg_launch_ms']*1e3))"
  timeout 600 python -m pytest tests/test_gpu_small.py -q -x -k "two_sweeps" 2>&1 | tail -1
  ( cd /tmp; rm -rf /tmp/q_f /tmp/q_w
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/q_f -o r -- python $R/tools/bench_configs.py "${args[@]}" --reps 1 > /dev/null 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/q_w -o r -- python $R/tools/bench_configs.py "${args[@]}" --reps 1 > /dev/null 2>&1
    for d in /tmp/q_f /tmp/q_w; do python $R/tools/prof_summary.py counters $(db $d) | python -c "
import sys
L = [l for l in sys.stdin if '_SIZE' in l]
l = max(L, key=lambda l: float(l[92:108]))
print(l[:40].strip(), l[65:83].strip(), 'n', l[84:91].strip(), 'mean KiB', l[92:108].strip())"; done )
done
