#!/bin/bash
# rocprofv3 kernel stats and HBM byte counters (separate --pmc passes) of the space transfers: tools/transfer_bench.py
# on the GPU box from the repository root; summaries -> gpurun_out/transfer_prof/
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/transfer_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/transfer_bench.py > $OUT/trace.log 2>&1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
for PMC in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/$PMC -- python3 $ROOT/tools/transfer_bench.py > $OUT/$PMC.log 2>&1 || echo "pass $PMC failed"
  find $OUT/$PMC -name "*counter_collection.csv" -exec cp {} $OUT/$PMC.csv \;
done
python3 - <<PY
import csv, collections
out = open("$OUT/summary.txt", "w")
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    try:
        for row in csv.DictReader(open("$OUT/%s.csv" % c)):
            k = row["Kernel_Name"]
            if "cell_" in k or "axis_apply" in k:
                acc[k[:110]].append(float(row["Counter_Value"]))
    except OSError as e:
        print(c, "missing", e, file=out)
    for k, v in sorted(acc.items()):
        print("%s %s: mean %.1f KB per launch over %d launches, max %.1f" % (c, k, sum(v) / len(v), len(v), max(v)), file=out)
PY
cat $OUT/summary.txt
