#!/bin/bash
# usage: tools/lzscan.sh <waves> lz...
W=$1; shift
for lz in "$@"; do
  echo -n "WAVES=$W LZ=$lz "; STFEM_TILE_WAVES=$W STFEM_TILE_LZ=$lz python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '"kernel_ms": [0-9.]*'
done
