#!/bin/bash
# The reference's space-time convergence study (tests/tp_01.cc, space_time_conv_test; its 2D output: tests/tp_01.output) in 3D on the device:
# heat and wave equation, FGMRES + space-time multigrid, refinements 1.. (cells 8, 64, 512, 4096), tau = 2^-(r+1), two steps per slab.
# Output -> gpurun_out/convergence.txt: cells s-dofs t-dofs Linf-Linf L2-L2 L2-H1semi iterations
cd dealii-stfem_amd/host || exit 1
out=../../gpurun_out/convergence.txt
: > $out
for prob in heat wave; do
  for cfg in "0 1 4" "1 1 4" "0 2 3" "1 2 3"; do
    set -- $cfg
    echo "== $prob type=$1 (0 cG, 1 dG) k=$2, Q$(( $2 + 1 ))" >> $out
    for r in $(seq 1 $3); do timeout -k 10 600 ./${prob}_convergence $1 $2 $r 2 mg=1 mg_float=1 2>/dev/null >> $out; done
  done
done
