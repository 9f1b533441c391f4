#!/usr/bin/env python3
"""Timing of the Stokes two-field space-time vmult (first version of the kernel):
Q2/Q1 x cG(1), unit cube, N^3 cells.  Prints DoF/s and algorithmic GB/s (16 B per DoF per vmult)."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
stfem = importlib.import_module("dealii-stfem_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
r = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dg = len(sys.argv) > 3 and sys.argv[3] == "dg"  # FE_DGP(1) pressure
op = stfem.StokesMatrixFreeOperator((N, N, N), viscosity=1.0, dg_pressure=dg)
Alpha, Beta, _, _ = stfem.get_fe_time_weights_stokes(stfem.CGP, r, 1.0 / 64, 1)
nt = r
rng = np.random.default_rng(0)
src, dst = [None] * (2 * nt), [None] * (2 * nt)
for d in range(nt):
    for v in range(2):
        j = stfem.stokes_block_index(nt, 0, v, d)
        n = 3 * op.n_velocity if v == 0 else op.n_pressure
        src[j] = op.initialize_dof_vector(v, rng.uniform(-1, 1, n))
        dst[j] = op.initialize_dof_vector(v)
for _ in range(3):
    op.st_vmult(Alpha, Beta, 1, nt, dst, src)
dst[0].download()
reps = int(os.environ.get('STOKES_BENCH_REPS', '100'))
t0 = time.perf_counter()
for _ in range(reps):
    op.st_vmult(Alpha, Beta, 1, nt, dst, src)
dst[0].download()  # synchronises (includes one device-to-host copy of a velocity block)
t1 = time.perf_counter()
dl0 = time.perf_counter(); dst[0].download(); dl = time.perf_counter() - dl0
ms = ((t1 - t0) - dl) / reps * 1e3
dofs = nt * (3 * op.n_velocity + op.n_pressure)
print(f"Stokes Q2/{'P1disc' if dg else 'Q1'} x cG({r}), {N}^3 cells, {dofs} space-time DoFs: {ms:.3f} ms per vmult, "
      f"{dofs / ms * 1e3:.3e} DoF/s, {16 * dofs / ms * 1e-6:.1f} GB/s algorithmic")
