#!/bin/bash
# Stokes Kronecker path: pressure gradient added inside the velocity sweep (default) against the separate gradient kernel (STFEM_STOKES_GRAD_KERNEL=1)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2 3; do
  for s in 0 1; do
    export STFEM_STOKES_GRAD_KERNEL=$s
    echo "grad_kernel=$s: $(python3 tools/stokes_bench.py 64 1 | tail -1 | cut -c60-110) | $(python3 tools/stokes_bench.py 96 1 | tail -1 | cut -c60-110) | $(python3 tools/stokes_bench.py 128 1 | tail -1 | cut -c60-112)"
  done
done
