#!/bin/bash
# tools/pmc.sh <tag> "<counters pass 1>" "<counters pass 2>" ...   (each pass = one rocprofv3 --pmc run)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/p$i -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline $BENCH_ARGS > $OUT/p$i.log 2>&1 || echo "pass $i failed"
  find $OUT/p$i -name "*counter_collection.csv" -exec cp {} $OUT/pmc$i.csv \;
done
python3 $ROOT/tools/summarize_pmc.py $OUT | grep -v "^=="
