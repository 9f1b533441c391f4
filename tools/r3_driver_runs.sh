#!/bin/bash
# Round 3: heat driver with the space-time multigrid on the cfg-1 mesh (Q4 x cG(2), fp32 levels, two slabs) - ms per FGMRES iteration,
# and a rocprofv3 kernel + marker trace of the same run.  Output: gpurun_out/r3_driver.txt, gpurun_out/r3_driver_prof/
cd ${GRAFT_REPO_ROOT:-$(pwd)}/dealii-stfem_amd/host || exit 1
out=../../gpurun_out/r3_driver.txt
: > $out
run() { echo "== heat_convergence $*" >> $out; timeout -k 10 400 ./heat_convergence "$@" >> $out 2>&1; echo "rc=$?" >> $out; }
run 0 2 6 1 2 0.5 4 72 0.015625 mg=1 mg_float=1
run 0 2 6 1 2 0.5 4 72 0.015625 mg=1
run 0 2 5 1 2 0.5 4 32 0.03125 mg=1 mg_float=1
cd /tmp && export TMPDIR=/tmp
D=${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 500 rocprofv3 --kernel-trace --marker-trace --stats --output-format csv -d $D/gpurun_out/r3_driver_prof -- $D/dealii-stfem_amd/host/heat_convergence 0 2 6 1 2 0.5 4 72 0.015625 mg=1 mg_float=1 > $D/gpurun_out/r3_driver_prof.log 2>&1
find $D/gpurun_out/r3_driver_prof -name "*stats.csv" | head
tail -5 $D/gpurun_out/r3_driver.txt
