#!/bin/bash
# Usage (on the GPU box, from the repo root):  tools/profile.sh <tag> [bench args...]
# Writes rocprofv3 kernel-trace stats and PMC passes under gpurun_out/prof_<tag>/ and copies the
# summaries to gpurun_out/profiles_<tag>/ (copy those into profiles/ to commit them).
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
SUM=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT $SUM
cd /tmp && export TMPDIR=/tmp
ARGS="--steps ${STEPS:-60} --warmup ${WARMUP:-10} --no-cpu-baseline $@"  # >= 50 back-to-back launches: the kernel average is the throttled steady state
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $SUM/kernel_stats.csv \;
# separate counter passes (never combined with tracing domains other than kernel-trace)
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc$i -- python3 $ROOT/bench.py $ARGS > $OUT/pmc$i.log 2>&1 || echo "pmc pass $i failed: $PMC"
  find $OUT/pmc$i -name "*counter_collection.csv" -exec cp {} $SUM/pmc$i.csv \;
done
python3 $ROOT/tools/summarize_pmc.py $SUM > $SUM/summary.txt 2>&1 || true
cat $SUM/summary.txt
