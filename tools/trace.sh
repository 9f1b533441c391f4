#!/bin/bash
# tools/trace.sh <tag> [bench args]: rocprofv3 kernel-trace stats only (fast)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $OUT/log.txt 2>&1
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
python3 - <<PY
import csv
for row in csv.DictReader(open("$OUT/kernel_stats.csv")):
    n=row["Name"]
    if "stfem" in n: print(n[:80], row["Calls"], row["AverageNs"], row["MinNs"], row["MaxNs"])
PY
grep -o '"ms_per_step": [0-9.]*' $OUT/log.txt
