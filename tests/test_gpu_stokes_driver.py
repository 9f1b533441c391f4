"""Stokes slab driver (SURVEY 8 f-3 for BASELINE configs[4]): the reference's Stokes convergence test (tests/tp_03stokes.cc) in 3D on
the device - C++ caller host/stokes_convergence.cpp on host/stfem/stokes_solver.h: SystemMatrixStokes right-hand side, velocity load
vectors, FGMRES preconditioned by relaxation sweeps of the two-variable Vanka smoother, zero-mean pressure, error norms per variable -
against the dense direct-solve restatement oracle/slab_oracle.py::stokes_convergence_row_3d (the recipe tests/test_tp03stokes_reference.py
pins to the reference's own 2D tables)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "dealii-stfem_amd", "host")


@pytest.mark.parametrize("ttype,k,refinement,nu,dg", [
    (0, 1, 1, 1.0, 0),   # cG(1), 8 cells
    (1, 1, 1, 1.0, 0),   # dG(1)
    (0, 1, 2, 1.0, 0),   # cG(1), 64 cells
    (0, 2, 1, 0.1, 0),   # cG(2), smaller viscosity
    (0, 1, 2, 1.0, 1),   # FE_DGP(1) pressure (the reference's default, tests/json/stokes.json)
    (1, 1, 1, 1.0, 1),
])
def test_stokes_convergence_row(ttype, k, refinement, nu, dg):
    from oracle import slab_oracle
    exe = os.path.join(HOST, "stokes_convergence")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    res = subprocess.run([exe, str(ttype), str(k), str(refinement), "3", "0.4" if dg else "0.6", str(nu), f"dg={dg}"], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout + res.stderr
    cells, udofs, pdofs, tdofs, l8, l2, h1, l2p, its = res.stdout.split()
    n = 2 ** refinement
    assert int(cells) == n ** 3 and int(udofs) == 3 * (2 * n + 1) ** 3 and int(pdofs) == (4 * n ** 3 if dg else (n + 1) ** 3)
    want = np.array(slab_oracle.stokes_convergence_row_3d(ttype, k, refinement, nu, dg_pressure=bool(dg)))
    got = np.array([float(l8), float(l2), float(h1), float(l2p)])
    # the slab systems are solved to 1e-12 (relative): the error norms agree far below their own size
    assert np.allclose(got, want, rtol=1e-6, atol=1e-9), (got, want)
    assert float(its) < 400
