#!/usr/bin/env python3
"""Diagnostic: the class-based (MFMA) and the per-cell (HBM-streaming) Vanka apply on the same Cartesian mesh
(the second context gets a 1e-8 h vertex jitter so that it takes the general path), for a random and for a smooth
right-hand side, and the contraction of one preconditioned Richardson step.  usage: vanka_compare.py p n"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
stfem = importlib.import_module("dealii-stfem_amd")
p, n = int(sys.argv[1]), int(sys.argv[2])
nc = (n, n, n)
Alpha, Beta, _, _ = stfem.get_fe_time_weights(stfem.CGP, 2, 1.0 / 64, 1)
nb = Alpha.shape[0]
ctx = stfem.MatrixFreeOperator(p, nc)
A = stfem.SystemMatrix(ctx, Alpha, Beta)
V = stfem.PreconditionVanka(ctx, Alpha, Beta)
print("plan", V.plan, "classes", V.n_classes)
verts = stfem.mesh_vertices(nc, distort=1e-8, seed=1)
ctx2 = stfem.MatrixFreeOperator(p, nc, vertices=verts)
V2 = stfem.PreconditionVanka(ctx2, Alpha, Beta)
rng = np.random.default_rng(0)
nd = p * n + 1
x1 = np.sin(np.pi * np.arange(nd) / (nd - 1))
smooth = np.einsum("i,j,k->ijk", x1, x1, x1).ravel()
for name, X in (("random", rng.uniform(-1, 1, (nb, ctx.n_dofs))), ("smooth", np.stack([smooth, 0.5 * smooth]))):
    xs = stfem.BlockVector(ctx, nb).upload(X)
    r = A.initialize_dof_vector()
    A.vmult(r, xs)                       # a residual-like vector: zero on the constrained rows
    R = r.download()
    z = stfem.BlockVector(ctx, nb)
    V.vmult(z, r)
    Z1 = z.download()
    z2 = stfem.BlockVector(ctx2, nb)
    V2.vmult(z2, stfem.BlockVector(ctx2, nb).upload(R))
    Z2 = z2.download()
    w = A.initialize_dof_vector()
    A.vmult(w, z)
    W = w.download()
    print(f"{name}: |V r - V2 r| / |V2 r| = {np.linalg.norm(Z1 - Z2) / np.linalg.norm(Z2):.3e}   |r - A V r| / |r| = "
          f"{np.linalg.norm(R - W) / np.linalg.norm(R):.3e}   max |V r| = {np.abs(Z1).max():.3e}  max |V2 r| = {np.abs(Z2).max():.3e}")
    bad = np.abs(Z1 - Z2) > 1e-5 * np.abs(Z2).max()
    if bad.any():
        ii = np.argwhere(bad)[:, 1]
        print("  differing entries:", bad.sum(), "x", (ii % nd).min(), (ii % nd).max(), "y", ((ii // nd) % nd).min(), ((ii // nd) % nd).max(),
              "z", (ii // (nd * nd)).min(), (ii // (nd * nd)).max())
