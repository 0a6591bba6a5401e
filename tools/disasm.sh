#!/bin/bash
# disassemble the gfx950 code object of one translation unit:  bash tools/disasm.sh build/obj/xinv_tu_bih.o out.s
L=/opt/rocm/lib/llvm/bin
t=$(mktemp -d)
$L/llvm-objcopy --dump-section .hip_fatbin=$t/fat.bin "$1"
$L/clang-offload-bundler --unbundle --type=o --input=$t/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$t/dev.co
$L/llvm-objdump -d --no-show-raw-insn $t/dev.co | c++filt > "$2"
rm -rf $t
