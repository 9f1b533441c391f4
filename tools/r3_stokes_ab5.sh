#!/bin/bash
# Stokes Kronecker path: divergence kernel as a march along z (default) against the gather form (STFEM_STOKES_DIV_GATHER=1)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2 3; do
  for s in 0 1; do
    export STFEM_STOKES_DIV_GATHER=$s
    echo "div_gather=$s: $(python3 tools/stokes_bench.py 64 1 | tail -1 | cut -c60-110) | $(python3 tools/stokes_bench.py 96 1 | tail -1 | cut -c60-110) | $(python3 tools/stokes_bench.py 64 2 | tail -1 | cut -c60-112) | serial $(STFEM_STOKES_SERIAL=1 python3 tools/stokes_bench.py 64 1 | tail -1 | cut -c60-80)"
  done
done
