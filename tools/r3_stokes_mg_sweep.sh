#!/bin/bash
# FGMRES iterations per slab of the Stokes driver against the smoother's parameters (8^3 cells, cG(1), two slabs)
cd ${GRAFT_REPO_ROOT:-$(pwd)}/dealii-stfem_amd/host
for sw in 1 2 3; do for om in 0.3 0.5 0.7 0.9 1.1; do
  echo "sweeps=$sw omega=$om mg=3: $(timeout -k 10 300 ./stokes_convergence 0 1 3 $sw $om 1.0 8 0.125 mg=3 2>&1 | awk '{print $NF}')   mg=2: $(timeout -k 10 300 ./stokes_convergence 0 1 3 $sw $om 1.0 8 0.125 mg=2 2>&1 | awk '{print $NF}')  none: $(timeout -k 10 300 ./stokes_convergence 0 1 3 $sw $om 1.0 8 0.125 2>&1 | awk '{print $NF}')"
done; done
