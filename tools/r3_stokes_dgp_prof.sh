#!/bin/bash
# Kernel stats of the Stokes vmult with the FE_DGP(1) pressure (64^3 cells, cG(1)) and of the FE_Q(5) x cG(2) heat vmult (tile sweep)
D=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$D/gpurun_out/prof_r3_dgp; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dgp -- python3 $D/tools/stokes_bench.py 64 1 dg > $OUT/dgp.log 2>&1 || exit 1
find $OUT/dgp -name "*kernel_stats.csv" -exec cp {} $D/gpurun_out/kernel_stats_stokes_dgp.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/q5 -- python3 $D/bench.py --degree 5 --cells 58 --no-cpu-baseline --steps 60 --warmup 10 > $OUT/q5.log 2>&1 || exit 1
find $OUT/q5 -name "*kernel_stats.csv" -exec cp {} $D/gpurun_out/kernel_stats_q5.csv \;
tail -1 $OUT/dgp.log; cut -c1-200 $D/gpurun_out/kernel_stats_stokes_dgp.csv | head -6; cut -c1-200 $D/gpurun_out/kernel_stats_q5.csv | head -5
