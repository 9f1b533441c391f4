#!/bin/bash
for pad in 0 2 4 8; do
  echo -n "pad=$pad "; STFEM_LIB=$PWD/dealii-stfem_amd/libstfem_pad$pad.so python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '"kernel_ms": [0-9.]*'
done
echo -n "prio exp "; STFEM_EXP=512 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '"kernel_ms": [0-9.]*'
echo -n "base     "; python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '"kernel_ms": [0-9.]*'
