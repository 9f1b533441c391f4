#!/bin/bash
# tools/vanka_pmc.sh <tag> [bench args]: rocprofv3 kernel stats + PMC passes of tools/vanka_bench.py (on the GPU box)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/vanka_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/vanka_bench.py "$@" > $OUT/trace.log 2>&1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc$i -- python3 $ROOT/tools/vanka_bench.py "$@" > $OUT/pmc$i.log 2>&1 || echo "pmc pass $i failed"
  find $OUT/pmc$i -name "*counter_collection.csv" -exec cp {} $OUT/pmc$i.csv \;
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
for f in sorted(glob.glob(os.path.join(d, "pmc*.csv"))):
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "vanka" in k:
            acc[k[:60]][row.get("Counter_Name")].append(float(row.get("Counter_Value", 0)))
    for k, cs in acc.items():
        print(k, {c: f"{sum(v) / len(v):.4g}" for c, v in cs.items()}, "launches", len(next(iter(cs.values()))))
PY
cat $OUT/kernel_stats.csv | cut -c1-200 | head -4
