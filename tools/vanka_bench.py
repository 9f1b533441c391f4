#!/usr/bin/env python3
"""Timing of the cell-patch Vanka smoother apply (stfem_vanka_vmult) on the cfg-1 mesh:
Q_p x cG(r), N^3 cells.  Prints the time per apply, cells/s and the MFMA rate (2 m^2 flop per cell,
m = n_blocks (p+1)^3) against the dense MFMA peak of the Number type (MI355X_MICROARCH.md: fp64 78.6,
fp32 157.3 TFLOP/s).  With a vertex jitter (5th argument, in units of h) the mesh is general and every cell has
its own block: the apply streams the blocks from HBM (algorithmic bytes = m^2 elements per cell, as in the reference)
and is reported against the 8 TB/s roofline.  usage: vanka_bench.py [N=72] [p=4] [r=2] [double|float] [distort=0]"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
stfem = importlib.import_module("dealii-stfem_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 72
p = int(sys.argv[2]) if len(sys.argv) > 2 else 4
r = int(sys.argv[3]) if len(sys.argv) > 3 else 2
number = sys.argv[4] if len(sys.argv) > 4 else "double"
distort = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
ctx = (stfem.MatrixFreeOperator(p, (N, N, N), vertices=stfem.mesh_vertices((N, N, N), distort=distort, seed=5489), number=number)
       if distort else stfem.MatrixFreeOperator(p, (N, N, N), number=number))
Alpha, Beta, _, _ = stfem.get_fe_time_weights(stfem.CGP, r, 1.0 / 144, 1)
nb = Alpha.shape[0]
t0 = time.perf_counter()
V = stfem.PreconditionVanka(ctx, Alpha, Beta)
setup = time.perf_counter() - t0
rng = np.random.default_rng(0)
src = stfem.BlockVector(ctx, nb).upload(rng.uniform(-1, 1, (nb, ctx.n_dofs)))
dst = stfem.BlockVector(ctx, nb)
for _ in range(3):
    V.vmult(dst, src)
stfem.dot(ctx, dst, dst)  # synchronises
reps = 20
t0 = time.perf_counter()
for _ in range(reps):
    V.vmult(dst, src)
stfem.dot(ctx, dst, dst)
ms = (time.perf_counter() - t0) / reps * 1e3
m = nb * (p + 1) ** 3
cells = N ** 3
flop = 2.0 * m * m * cells
peak = 78.6 if number == "double" else 157.3
if distort:
    es = 8 if number == "double" else 4
    gb = cells * float(m) * m * es * 1e-9
    print(f"Vanka apply Q{p} x cG({r}), per-cell {m} x {m} blocks, {N}^3 perturbed cells, {number}: set-up {setup:.1f} s, "
          f"{ms:.3f} ms per apply, {cells / ms * 1e3:.3e} cells/s, {gb:.2f} GB of blocks -> {gb / ms * 1e3:.0f} GB/s = "
          f"{gb / ms * 1e3 / 8000:.3f} of the 8 TB/s HBM roofline")
    sys.exit(0)
print(f"Vanka apply Q{p} x cG({r}) ({m} x {m} blocks, {V.n_classes} classes, plan {V.plan}), {N}^3 cells, {number}: set-up {setup:.2f} s, "
      f"{ms:.3f} ms per apply, {cells / ms * 1e3:.3e} cells/s, {flop / ms * 1e-9:.1f} TFLOP/s = "
      f"{flop / ms * 1e-9 / peak:.3f} of the {peak} TFLOP/s dense MFMA peak; DoF traffic {2 * nb * ctx.n_dofs * (8 if number == 'double' else 4) / ms * 1e-6:.0f} GB/s")
