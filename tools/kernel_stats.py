#!/usr/bin/env python3
"""Registers, LDS, scratch and spill counts of the kernels in one object file's gfx950 code object.
  python tools/kernel_stats.py build/obj/xinv_tu_fused3d.o [substring]"""
import re, subprocess, sys, tempfile
LLVM = '/opt/rocm/lib/llvm/bin'
o = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ''
with tempfile.TemporaryDirectory() as td:
    co = td + '/dev.co'; fat = td + '/fat.bin'
    subprocess.run([LLVM + '/llvm-objcopy', '--dump-section', '.hip_fatbin=' + fat, o], check=True)
    subprocess.run([LLVM + '/clang-offload-bundler', '--unbundle', '--type=o', '--input=' + fat,
                    '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + co], check=True)
    txt = subprocess.run([LLVM + '/llvm-readelf', '--notes', co], capture_output=True, text=True).stdout
for blk in txt.split('- .agpr_count:')[1:]:
    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    if pat not in dem:
        continue
    g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, blk).group(1)
    print('vgpr %3s sgpr %3s scratch %4s lds %6s vspill %3s sspill %3s  %s' % (
        g('vgpr_count'), g('sgpr_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size'),
        g('vgpr_spill_count'), g('sgpr_spill_count'), dem[:110]))
