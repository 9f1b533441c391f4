#!/bin/bash
cd /tmp && export TMPDIR=/tmp
D=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats --output-format csv -d $D/gpurun_out/r3_stokes_prof2 -- python3 $D/tools/stokes_bench.py 64 1 > $D/gpurun_out/r3_stokes_prof2.log 2>&1
find $D/gpurun_out/r3_stokes_prof2 -name "*kernel_stats.csv" -exec head -6 {} \; | cut -c1-200
export STFEM_STOKES_SERIAL=1
rocprofv3 --kernel-trace --stats --output-format csv -d $D/gpurun_out/r3_stokes_prof3 -- python3 $D/tools/stokes_bench.py 64 1 > $D/gpurun_out/r3_stokes_prof3.log 2>&1
find $D/gpurun_out/r3_stokes_prof3 -name "*kernel_stats.csv" -exec head -6 {} \; | cut -c1-200
