#!/bin/bash
# kdis.sh <object file> [out.s]: disassembly of the gfx950 code object inside a hipcc object (csi-nn2_amd/lib/obj/*.o)
LL=/opt/rocm/lib/llvm/bin
d=$(mktemp -d)
$LL/llvm-objcopy --dump-section .hip_fatbin=$d/fat.bin "$1"
$LL/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$d/fat.bin --output=$d/k.co --unbundle
$LL/llvm-objdump -d $d/k.co > "${2:-/dev/stdout}"
rm -rf $d
