#!/bin/bash
# Heat driver with the space-time multigrid as preconditioner against the Vanka-relaxation-only runs of
# profiles/r2/driver.txt.  Run on the GPU box from the repository root; output -> gpurun_out/stmg_runs.txt
cd dealii-stfem_amd/host || exit 1
out=../../gpurun_out/stmg_runs.txt
: > $out
run() { echo "== heat_convergence $*" >> $out; timeout -k 10 400 ./heat_convergence "$@" >> $out 2>&1; echo "rc=$?" >> $out; }
if [ "$1" != "q4" ]; then
for r in 1 2 3 4; do run 0 1 $r 2 mg=1; done
run 0 1 4 2 2
run 1 1 3 2 mg=1
run 1 1 3 2 mg=1 mg_float=1
fi
# Q4 x cG(2), one time step per slab (the Vanka blocks hold at most three temporal blocks of Q4), tau = 1/64, two slabs
for n in 12 16 32; do run 0 2 5 1 2 0.5 4 $n 0.03125 mg=1; done
run 0 2 5 1 2 0.5 4 32 0.03125 mg=1 mg_float=1
run 0 2 5 1 2 0.5 4 32 0.03125 mg=1 pmg=1 coarsening=space_and_time
# the cfg-1 mesh, tau = 2^-7, two slabs
run 0 2 6 1 2 0.5 4 72 0.015625 mg=1
run 0 2 6 1 2 0.5 4 72 0.015625 mg=1 mg_float=1
