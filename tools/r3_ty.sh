#!/bin/bash
# Q2 pencils with two (default) and four cell rows: configs[0]-type system and the Stokes Kronecker path
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for ty in ${TYS:-2 4}; do
  export STFEM_PENCIL_TY=$ty
  echo "== STFEM_PENCIL_TY=$ty"
  python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --degree 2 --time-degree 1 --timesteps-at-once 4 --cells 144 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('Q2 cG(1)x4 144^3:', round(d['ms_per_step'],4), 'ms', round(d['roofline']['frac'],4))"
  python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --degree 2 --time-degree 2 --cells 144 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('Q2 cG(2) 144^3:', round(d['ms_per_step'],4), 'ms', round(d['roofline']['frac'],4))"
  python3 tools/stokes_bench.py 64 1
  python3 tools/stokes_bench.py 96 1
done
