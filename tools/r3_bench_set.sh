#!/bin/bash
# Round-3 bench lines on one GPU box: headline, many-block systems, general path, Stokes.  Output: gpurun_out/r3_bench_<tag>.json
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
python3 bench.py --steps 50 --warmup 5 > gpurun_out/r3_bench_cfg1.json 2> gpurun_out/r3_bench_cfg1.err
python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --timesteps-at-once 2 > gpurun_out/r3_bench_q4_cg2x2.json 2>&1
python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --degree 2 --time-degree 1 --timesteps-at-once 4 --cells 144 > gpurun_out/r3_bench_q2_cg1x4.json 2>&1
python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --config 3 > gpurun_out/r3_bench_cfg3.json 2>&1
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --distort 0.15 > gpurun_out/r3_bench_general.json 2>&1
tail -n 1 gpurun_out/r3_bench_*.json | cut -c1-400
