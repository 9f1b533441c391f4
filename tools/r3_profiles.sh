#!/bin/bash
# Round-3 profile set on one GPU box: cfg 1 (headline), cfg 3, general path, Stokes (Kronecker path).  Summaries: gpurun_out/profiles_<tag>/
D=${GRAFT_REPO_ROOT:-$(pwd)}
cd $D
bash tools/profile.sh r3_cfg1 > gpurun_out/r3_prof_cfg1.log 2>&1 && echo "cfg1 done" &&
bash tools/profile.sh r3_cfg3 --config 3 > gpurun_out/r3_prof_cfg3.log 2>&1 && echo "cfg3 done" &&
STEPS=20 WARMUP=3 bash tools/profile.sh r3_general --distort 0.15 > gpurun_out/r3_prof_general.log 2>&1 && echo "general done" || exit 1
# Stokes: kernel stats + counter passes of tools/stokes_bench.py (64^3 cells, cG(1))
OUT=$D/gpurun_out/prof_r3_stokes; SUM=$D/gpurun_out/profiles_r3_stokes; mkdir -p $OUT $SUM
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $D/tools/stokes_bench.py 64 1 > $OUT/trace.log 2>&1 || exit 1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $SUM/kernel_stats.csv \;
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc$i -- python3 $D/tools/stokes_bench.py 64 1 > $OUT/pmc$i.log 2>&1 || echo "pmc pass $i failed"
  find $OUT/pmc$i -name "*counter_collection.csv" -exec cp {} $SUM/pmc$i.csv \;
done
python3 $D/tools/summarize_pmc.py $SUM > $SUM/summary.txt 2>&1 || true
tail -20 $SUM/summary.txt | cut -c1-300
echo "stokes done"
