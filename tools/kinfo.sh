#!/bin/bash
# kinfo.sh <object.o> <name filter>: register / LDS / code size of the kernels in a HIP object
set -e
T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "$1" $T/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/k.elf
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/k.elf | grep -E "^\s+\.name:|vgpr_count|sgpr_spill|vgpr_spill|group_segment_fixed" | paste - - - - - | grep -E "${2:-.}" | sed -e 's/  */ /g' | cut -c1-260
/opt/rocm/lib/llvm/bin/llvm-readelf -sW $T/k.elf | grep FUNC | grep -E "${2:-.}" | awk '{print $3, $8}' | sort -u | cut -c1-200
if [ -n "$3" ]; then /opt/rocm/lib/llvm/bin/llvm-objdump -d $T/k.elf > "$3"; fi
rm -rf $T
