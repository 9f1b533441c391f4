"""oracle/slab_oracle.py (3D heat convergence row, dense direct solves): the recipe itself is pinned to the reference's
2D numbers by tests/test_tp01_reference.py; here the 3D version is checked for what it must show - the convergence
orders of FE_Q(k + 1) x cG(k) / dG(k) under simultaneous refinement (tests/tp_01.output shows the same in 2D)."""
import numpy as np

from oracle import slab_oracle


def test_convergence_orders_3d():
    # cG(1) with Q2: L2(L2) error ~ tau^2 + h^3 -> halves by about 4 (tau also halves); dG(0) with Q1: first order in time
    e1 = slab_oracle.heat_convergence_row_3d(0, 1, 2, 2)
    e2 = slab_oracle.heat_convergence_row_3d(0, 1, 3, 2)
    assert 3.0 < e1[1] / e2[1] < 8.0
    d1 = slab_oracle.heat_convergence_row_3d(1, 0, 2, 2)
    d2 = slab_oracle.heat_convergence_row_3d(1, 0, 3, 2)
    assert 1.3 < d1[1] / d2[1] < 3.0
    assert all(np.isfinite(v) for v in e1 + e2 + d1 + d2)
