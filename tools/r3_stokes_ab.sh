#!/bin/bash
# Stokes Kronecker path: divergence kernel beside the velocity sweep (default) against after it (STFEM_STOKES_SERIAL=1), same box
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
  for s in 0 1; do
    export STFEM_STOKES_SERIAL=$s
    echo "serial=$s: $(python3 tools/stokes_bench.py 64 1 | tail -1)"
    echo "serial=$s: $(python3 tools/stokes_bench.py 64 2 | tail -1)"
    echo "serial=$s: $(python3 tools/stokes_bench.py 96 1 | tail -1)"
  done
done
