#!/bin/bash
# Stokes: final bench lines and profile of the round (after the gradient kernel change)
D=${GRAFT_REPO_ROOT:-$(pwd)}
cd $D
python3 bench.py --steps 100 --warmup 5 --config 4 > gpurun_out/r3_final_cfg4_stokes.json 2>&1
python3 bench.py --steps 100 --warmup 5 --config 4 --cells 96 > gpurun_out/r3_final_cfg4_stokes_96.json 2>&1
python3 bench.py --steps 100 --warmup 5 --config 4 --time-degree 3 > gpurun_out/r3_final_cfg4_stokes_cg3.json 2>&1
OUT=$D/gpurun_out/prof_r3_stokes; SUM=$D/gpurun_out/profiles_r3_stokes; rm -rf $OUT $SUM; mkdir -p $OUT $SUM
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
head -6 $SUM/summary.txt | cut -c1-200
cd $D
for f in gpurun_out/r3_final_cfg4*.json; do python3 -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1])
print('$f'.split('r3_final_')[1], 'ms', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],4))
"; done
