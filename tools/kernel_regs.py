#!/usr/bin/env python3
"""VGPR / SGPR / scratch use of the kernels in libxinv_hip.so (from the code objects' metadata notes).
  python tools/kernel_regs.py [substring ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
def main():
    pats = sys.argv[1:]
    objs = [os.path.join(ROOT, 'build', 'obj', f) for f in sorted(os.listdir(os.path.join(ROOT, 'build', 'obj'))) if f.endswith('.o')]
    for o in objs:
        with tempfile.TemporaryDirectory() as td:
            co = os.path.join(td, 'dev.co'); fat = os.path.join(td, 'fat.bin')
            subprocess.run([LLVM + '/llvm-objcopy', '--dump-section', '.hip_fatbin=' + fat, o], capture_output=True)
            if not os.path.exists(fat):
                continue
            r = subprocess.run([LLVM + '/clang-offload-bundler', '--unbundle', '--type=o', '--input=' + fat,
                                '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + co], capture_output=True, text=True)
            if r.returncode or not os.path.exists(co):
                continue
            txt = subprocess.run([LLVM + '/llvm-readelf', '--notes', co], capture_output=True, text=True).stdout
            for blk in txt.split('- .agpr_count:')[1:]:
                name = re.search(r'\.name:\s+(\S+)', blk)
                vg = re.search(r'\.vgpr_count:\s+(\d+)', blk); sg = re.search(r'\.sgpr_count:\s+(\d+)', blk)
                sc = re.search(r'\.private_segment_fixed_size:\s+(\d+)', blk); lds = re.search(r'\.group_segment_fixed_size:\s+(\d+)', blk)
                if not name: continue
                dem = subprocess.run(['c++filt', name.group(1)], capture_output=True, text=True).stdout.strip()
                if pats and not all(p in dem for p in pats): continue
                print('%4s vgpr %4s sgpr %5s scratch %6s lds  %s' % (vg.group(1), sg.group(1), sc.group(1), lds.group(1), dem[:150]))
main()
