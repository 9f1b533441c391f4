#!/bin/bash
echo -n "1 WG/CU  "; STFEM_LIB=$PWD/dealii-stfem_amd/libstfem_occ1.so python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '"kernel_ms": [0-9.]*'
echo -n "2 WG/CU  "; STFEM_TILE_WAVES=2 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '"kernel_ms": [0-9.]*'
echo -n "3 WG/CU  "; STFEM_TILE_WAVES=3 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '"kernel_ms": [0-9.]*'
for e in 13 269; do echo -n "1 WG/CU EXP=$e "; STFEM_EXP=$e STFEM_LIB=$PWD/dealii-stfem_amd/libstfem_occ1.so python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '"kernel_ms": [0-9.]*'; done
for e in 13 269; do echo -n "3 WG/CU EXP=$e "; STFEM_EXP=$e STFEM_TILE_WAVES=3 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '"kernel_ms": [0-9.]*'; done
