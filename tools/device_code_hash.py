#!/usr/bin/env python3
"""sha256 of the gfx950 .text section of every object of the shipped build (build/obj/*.o): two trees whose hashes agree
compile to the same device code (a source clean-up that must not change a kernel).   python tools/device_code_hash.py"""
import hashlib, os, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
for f in sorted(os.listdir(os.path.join(ROOT, 'build', 'obj'))):
    if not f.endswith('.o'):
        continue
    o = os.path.join(ROOT, 'build', 'obj', f)
    with tempfile.TemporaryDirectory() as td:
        fat, co, txt = (os.path.join(td, x) for x in ('fat', 'co', 'text'))
        subprocess.run([LLVM + '/llvm-objcopy', '--dump-section', '.hip_fatbin=' + fat, o], capture_output=True)
        if not os.path.exists(fat):
            continue
        subprocess.run([LLVM + '/clang-offload-bundler', '--unbundle', '--type=o', '--input=' + fat,
                        '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + co], capture_output=True)
        subprocess.run([LLVM + '/llvm-objcopy', '--dump-section', '.text=' + txt, co], capture_output=True)
        print(f, hashlib.sha256(open(txt, 'rb').read()).hexdigest()[:16] if os.path.exists(txt) else 'no text')
