#!/bin/bash
# FE_DGP(1) Stokes vmult, A/B of two library builds on one box: dealii-stfem_amd/libstfem_hip_old.so (before) against the tree's library
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
  for a in "64 1 dg" "96 1 dg"; do
    echo "old: $(STFEM_LIB=$PWD/dealii-stfem_amd/libstfem_hip_old.so python tools/stokes_bench.py $a)"
    echo "new: $(python tools/stokes_bench.py $a)"
  done
done
