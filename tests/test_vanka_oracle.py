"""Properties of the numpy restatement of PreconditionVanka (oracle/vanka_oracle.py, reference
include/stmg.h:619-907).  The reference holds no vector of the smoother, so the restatement is checked by
what it must satisfy: on a one-cell mesh the patch IS the system (exact inverse), the valence counts cells,
constrained rows keep only their diagonal."""
import importlib

import numpy as np

from oracle import oracle, vanka_oracle


def test_one_cell_mesh_is_the_exact_inverse():
    stfem = importlib.import_module("dealii-stfem_amd")
    p, nc = 2, (1, 1, 1)
    Alpha, Beta, _, _ = oracle.time_weights(0, 2, 0.1, 1)
    V = vanka_oracle.VankaOracle(p, nc, stfem.mesh_vertices(nc), 63, Alpha, Beta)
    A = np.kron(Alpha, V.K) + np.kron(Beta, V.M)  # the assembled, constrained space-time matrix
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, (Alpha.shape[0], V.N))
    y = V.vmult((A @ x.ravel()).reshape(x.shape))
    assert np.allclose(y, x, rtol=0, atol=1e-10)
    assert np.all(V.valence == 1.0)


def test_assembled_matrices_and_valence():
    stfem = importlib.import_module("dealii-stfem_amd")
    p, nc = 2, (3, 2, 2)
    Alpha, Beta, _, _ = oracle.time_weights(0, 1, 0.1, 1)
    V = vanka_oracle.VankaOracle(p, nc, stfem.mesh_vertices(nc), 63, Alpha, Beta)
    nd = V.nd
    val = V.valence.reshape(nd[::-1])
    assert val.max() == 8.0 and val.min() == 1.0
    # [z][y][x]: an interior vertex belongs to eight cells, a node on an interior edge to four, a cell-interior node to one
    assert val[2, 2, 2] == 8.0 and val[2, 2, 1] == 4.0 and val[2, 1, 1] == 2.0 and val[1, 1, 1] == 1.0
    con = V.constrained
    # constrained rows / columns: only the (positive) diagonal is left; the free part is the constrained operator
    offK = V.K - np.diag(V.K.diagonal())
    assert np.all(offK[con, :] == 0) and np.all(offK[:, con] == 0) and np.all(V.K.diagonal()[con] > 0)
    ref = oracle.Oracle(p, nc, stfem.mesh_vertices(nc), 63).dense(laplace=1.0)
    free = ~con
    assert np.allclose(V.K[np.ix_(free, free)], ref[np.ix_(free, free)], rtol=0, atol=1e-12)
    # an additive-Schwarz sweep with weighted blocks reproduces constants of the partition: sum_c R_c^T D_c^-1 R_c = I
    # for the block-diagonal part: with Alpha = 0, Beta = 1 and M replaced by its diagonal the smoother is M_diag^-1
    x = np.random.default_rng(1).uniform(-1, 1, (Alpha.shape[0], V.N))
    assert np.all(np.isfinite(V.vmult(x)))
