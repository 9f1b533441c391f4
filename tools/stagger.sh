#!/bin/bash
for s in ${@:-0 2 4 6 8 12}; do
  echo -n "STFEM_STAGGER=$s "; STFEM_STAGGER=$s python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '"kernel_ms": [0-9.]*'
done
