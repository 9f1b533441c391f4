#!/bin/bash
# Stokes slab driver (host/stokes_convergence): iterations and wall time per slab, Vanka sweeps alone against the multigrid
cd ${GRAFT_REPO_ROOT:-$(pwd)}/dealii-stfem_amd/host || exit 1
out=../../gpurun_out/r3_stokes_driver.txt
: > $out
run() { echo "== stokes_convergence $*" >> $out; timeout -k 10 500 ./stokes_convergence "$@" >> $out 2>&1; echo "rc=$?" >> $out; }
run 0 1 3 2 0 1.0 8 0.125
run 0 1 3 2 0 1.0 8 0.125 mg=3
run 0 1 4 2 0 1.0 16 0.0625 mg=4
run 0 1 5 2 0 1.0 32 0.03125 mg=5
run 0 2 4 2 0 1.0 16 0.0625 mg=4
run 0 1 6 2 0 1.0 64 0.015625 mg=6
cat $out
