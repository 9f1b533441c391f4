#!/bin/bash
# ablation of the tile kernel phases (results invalid, timing only); needs libstfem_abl.so built with
#   make -C dealii-stfem_amd/csrc stfem_tile_p4.o stfem_tile.o KFLAGS=-DSTFEM_ABLATION -B
for e in ${@:-0 1 2 4 8 14 15}; do
  echo -n "STFEM_EXP=$e "; STFEM_LIB=$PWD/dealii-stfem_amd/libstfem_abl.so STFEM_EXP=$e python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '"kernel_ms": [0-9.]*'
done
