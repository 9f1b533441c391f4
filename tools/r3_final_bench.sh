#!/bin/bash
# Final bench lines of round 3 on one GPU box.  Output: gpurun_out/r3_final_<tag>.json
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
python3 bench.py --steps 50 --warmup 5 > gpurun_out/r3_final_cfg1.json 2> gpurun_out/r3_final_cfg1.err
python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --timesteps-at-once 2 > gpurun_out/r3_final_q4_cg2x2.json 2>&1
python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --degree 2 --time-degree 1 --timesteps-at-once 4 --cells 144 > gpurun_out/r3_final_q2_cg1x4.json 2>&1
python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --config 3 > gpurun_out/r3_final_cfg3.json 2>&1
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --distort 0.15 > gpurun_out/r3_final_general.json 2>&1
python3 bench.py --steps 100 --warmup 5 --config 4 > gpurun_out/r3_final_cfg4_stokes.json 2>&1
python3 bench.py --steps 100 --warmup 5 --config 4 --cells 96 > gpurun_out/r3_final_cfg4_stokes_96.json 2>&1
for f in gpurun_out/r3_final_*.json; do python3 -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
print('$f'.split('r3_final_')[1], 'ms', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],4), 'traffic', d['roofline'].get('traffic'))
"; done
