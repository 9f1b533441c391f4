#!/bin/bash
# A/B of an experiment build of the pencil kernel against the product library on the same box:
#   tools/r3_ab.sh <name> [bench args]   (dealii-stfem_amd/libstfem_<name>.so from tools/build_pencil_exp.sh)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
N=$1; shift
for rep in 1 2 3; do
  for lib in product $N; do
    if [ $lib = product ]; then unset STFEM_LIB; else export STFEM_LIB=$PWD/dealii-stfem_amd/libstfem_$lib.so; fi
    python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$lib', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4))"
  done
done
