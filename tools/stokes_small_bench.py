#!/usr/bin/env python3
"""Cost of one Stokes vmult and one two-variable Vanka step on the small meshes of the multigrid's coarse levels (Q2/Q1 x cG(1)):
wall time per call of a back-to-back sequence (what the V-cycle pays), per mesh size.  STFEM_STOKES_SERIAL=1 in the environment puts
the divergence kernel on the caller's stream."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
stfem = importlib.import_module("dealii-stfem_amd")
dg = len(sys.argv) > 1 and sys.argv[1] == "dg"
Alpha, Beta, _, _ = stfem.get_fe_time_weights_stokes(stfem.CGP, 1, 1.0 / 64, 1)
for N in [int(a) for a in sys.argv[2:]] or (2, 4, 8, 16, 32):
    op = stfem.StokesMatrixFreeOperator((N, N, N), viscosity=1.0, dg_pressure=dg)
    rng = np.random.default_rng(0)
    src = [op.initialize_dof_vector(v, rng.uniform(-1, 1, 3 * op.n_velocity if v == 0 else op.n_pressure)) for v in (0, 1)]
    dst = [op.initialize_dof_vector(v) for v in (0, 1)]
    P = stfem.StokesPreconditionVanka(op, [0, 1], Alpha, Beta)

    def timed(f, reps=300):
        for _ in range(5):
            f()
        dst[1].download()
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        dst[1].download()
        return (time.perf_counter() - t0) / reps * 1e6

    tv = timed(lambda: op.st_vmult(Alpha, Beta, 1, 1, dst, src))
    tp = timed(lambda: P.step(dst, 0.4, True, src))
    print(f"{N:3d}^3 cells: vmult {tv:7.1f} us, Vanka step {tp:7.1f} us", flush=True)
    del P, op
