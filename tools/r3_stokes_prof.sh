#!/bin/bash
# rocprofv3 kernel stats of tools/stokes_bench.py (Kronecker path and, with STFEM_STOKES_CELL=1, the cell kernel)
cd /tmp && export TMPDIR=/tmp
D=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats --output-format csv -d $D/gpurun_out/r3_stokes_prof_cart -- python3 $D/tools/stokes_bench.py 64 1 > $D/gpurun_out/r3_stokes_prof_cart.log 2>&1
export STFEM_STOKES_CELL=1
rocprofv3 --kernel-trace --stats --output-format csv -d $D/gpurun_out/r3_stokes_prof_cell -- python3 $D/tools/stokes_bench.py 64 1 > $D/gpurun_out/r3_stokes_prof_cell.log 2>&1
for v in cart cell; do echo "== $v"; cat $D/gpurun_out/r3_stokes_prof_$v.log | tail -1; find $D/gpurun_out/r3_stokes_prof_$v -name "*kernel_stats.csv" -exec head -8 {} \; | cut -c1-170; done
