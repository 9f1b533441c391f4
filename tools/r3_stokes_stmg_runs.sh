#!/bin/bash
# Stokes slab driver with cG(2) in time: space levels only (mg=) against the reference's space-time sequence (stmg=1)
cd ${GRAFT_REPO_ROOT:-$(pwd)}/dealii-stfem_amd/host || exit 1
out=../../gpurun_out/r3_stokes_stmg.txt
: > $out
run() { echo "== stokes_convergence $*" >> $out; timeout -k 10 500 ./stokes_convergence "$@" >> $out 2>&1; echo "rc=$?" >> $out; }
run 0 2 3 2 0 1.0 8 0.125 mg=3
run 0 2 3 2 0 1.0 8 0.125 mg=3 stmg=1
run 0 2 3 2 0 1.0 8 0.125 mg=3 stmg=1 coarsening=space_and_time
run 0 2 4 2 0 1.0 16 0.0625 mg=4
run 0 2 4 2 0 1.0 16 0.0625 mg=4 stmg=1
run 1 1 3 2 0 1.0 8 0.125 mg=3 dg=1
run 1 1 3 2 0 1.0 8 0.125 mg=3 stmg=1 dg=1
grep -v relaxation $out
