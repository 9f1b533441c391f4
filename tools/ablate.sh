#!/bin/bash
# ablation of the tile kernel phases (results invalid, timing only)
for e in ${@:-0 1 2 4 8 14 15}; do
  echo -n "STFEM_EXP=$e "; STFEM_TILE_WAVES=2 STFEM_EXP=$e python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '"kernel_ms": [0-9.]*'
done
