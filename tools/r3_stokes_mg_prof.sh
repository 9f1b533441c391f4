#!/bin/bash
# Kernel trace of the multigrid-preconditioned Stokes slab solve (host/stokes_convergence, 32^3 cells, cG(1), five levels): where an FGMRES iteration goes
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/r3_stokes_mg_prof
rm -rf $out; mkdir -p $out
cd $root/dealii-stfem_amd/host || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/trace -o run --output-format csv -- ./stokes_convergence 0 1 ${STOKES_REF:-5} 2 0 1.0 ${STOKES_NC:-32} ${STOKES_DT:-0.03125} mg=${STOKES_REF:-5} ${STOKES_DG:+dg=1} > $out/run.txt 2>&1
echo "rc=$?" >> $out/run.txt
tail -5 $out/run.txt
f=$(find $out/trace -name '*kernel_stats.csv' | head -1)
python3 - "$f" > $out/kernel_stats_short.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms")
for r in rows[:40]:
    print(f"{r['Name'][:110]:110s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.1f} total_ms={float(r['TotalDurationNs'])/1e6:8.1f} pct={r['Percentage']}")
PY
cat $out/kernel_stats_short.txt
rm -rf $out/trace
