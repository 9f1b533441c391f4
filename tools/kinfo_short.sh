#!/bin/bash
# kinfo_short.sh <object.o> [filter]: one line per kernel: template args, VGPRs, VGPR spills, SGPR spills, scratch
T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "$1" $T/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/k.elf
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/k.elf | python3 -c '
import sys,re
cur={}
for line in sys.stdin:
    m=re.match(r"\s+\.(\w+):\s+(.*)",line)
    if not m: continue
    k,v=m.groups()
    if k=="name" and "kd" not in v:
        cur["name"]=v.strip()
    if k in ("vgpr_count","vgpr_spill_count","sgpr_spill_count","private_segment_fixed_size","agpr_count","sgpr_count"): cur[k]=v.strip()
    if k=="wavefront_size":
        n=cur.get("name","?")
        n=re.sub(r"_ZN5stfem3f\d+\d+_GLOBAL__N_1\d+","",n); n=re.sub(r"EvNS0_.*","",n)
        print(n, "vgpr",cur.get("vgpr_count"),"agpr",cur.get("agpr_count"),"vspill",cur.get("vgpr_spill_count"),"sspill",cur.get("sgpr_spill_count"),"scratch",cur.get("private_segment_fixed_size"))
        cur={}
' | grep -E "${2:-.}"
rm -rf $T
