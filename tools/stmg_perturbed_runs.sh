#!/bin/bash
# Heat driver on perturbed meshes (the configs[2] mesh type: general-geometry operator, one Vanka block per cell built on the device),
# FGMRES + space-time multigrid.  Output -> gpurun_out/stmg_perturbed.txt
cd dealii-stfem_amd/host || exit 1
out=../../gpurun_out/stmg_perturbed.txt
: > $out
run() { echo "== heat_convergence $*" >> $out; timeout -k 10 900 ./heat_convergence "$@" >> $out 2>&1; echo "rc=$?" >> $out; }
run 0 1 3 2 mg=1 distort=0.15
run 0 1 4 2 mg=1 distort=0.15
run 0 1 4 2 2 distort=0.15
run 0 2 5 1 2 0.5 4 32 0.03125 mg=1 mg_float=1 distort=0.15
run 0 2 6 1 2 0.5 4 72 0.015625 mg=1 mg_float=1 distort=0.15
