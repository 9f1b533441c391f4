#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for ty in 2 4; do
  export STFEM_PENCIL_TY=$ty
  python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --config 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('TY=$ty cfg3 Q3 dG(2) 80^3:', round(d['ms_per_step'],4), 'ms', round(d['roofline']['frac'],4))"
  python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --degree 3 --time-degree 2 --cells 96 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('TY=$ty Q3 cG(2) 96^3:', round(d['ms_per_step'],4), 'ms', round(d['roofline']['frac'],4))"
done
done
