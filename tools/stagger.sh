#!/bin/bash
# usage: tools/stagger.sh <div> s1 s2 ...
D=$1; shift
for s in "$@"; do
  echo -n "DIV=$D STAGGER=$s "; STFEM_STAGGER_DIV=$D STFEM_STAGGER=$s python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '"kernel_ms": [0-9.]*'
done
