"""CPU oracle of the Stokes two-field operator against the independent dense numpy fixtures
(tests/golden/make_golden_stokes.py).  SURVEY 8a-14."""
import os

import numpy as np
import pytest

from oracle import oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")
FIXTURES = ["stokes_cart_2x2x2", "stokes_pert_2x3x2", "stokes_free_3x2x2"]


def rel(a, b):
    return np.linalg.norm(np.ravel(a) - np.ravel(b)) / max(np.linalg.norm(np.ravel(b)), 1e-300)


@pytest.mark.parametrize("name", FIXTURES)
def test_stokes_oracle_matches_dense_fixture(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    orc = oracle.StokesOracle(tuple(g["ncell"]), g["vertices"], int(g["mask"]), float(g["nu"]))
    nt = int(g["nt"])
    for i in range(nt):
        ou, op = orc.apply(g["U"][i], g["P"][i])
        assert rel(ou, g["SU"][i]) < 1e-12 and rel(op, g["SP"][i]) < 1e-12
        mu, zero = orc.apply(g["U"][i], g["P"][i], 0.0, 1.0)
        assert rel(mu, g["MU"][i]) < 1e-12 and np.abs(zero).max() == 0.0
    blocks = [None] * (2 * nt)
    for d in range(nt):
        blocks[oracle.stokes_block_index(nt, 0, 0, d)] = g["U"][d].reshape(-1)
        blocks[oracle.stokes_block_index(nt, 0, 1, d)] = g["P"][d]
    dst = orc.st_vmult(g["Alpha"], g["Beta"], 1, nt, blocks)
    for d in range(nt):
        assert rel(dst[oracle.stokes_block_index(nt, 0, 0, d)], g["DU"][d]) < 1e-12
        assert rel(dst[oracle.stokes_block_index(nt, 0, 1, d)], g["DP"][d]) < 1e-12


def test_stokes_analytic_properties():
    """div of a constant field vanishes; constrained velocity rows stay zero; the viscous block is
    symmetric and annihilates rigid translations (no constraints)."""
    nc = (2, 2, 3)
    from tests.golden.make_golden import structured_vertices
    verts = structured_vertices(nc, (0, 0, 0), (1, 2, 1), 0.1, seed=3)
    orc = oracle.StokesOracle(nc, verts, 0, 0.7)
    U = np.stack([np.full(orc.n_u, 1.0), np.full(orc.n_u, -2.0), np.full(orc.n_u, 0.5)])
    ou, op = orc.apply(U, np.zeros(orc.n_p))
    assert np.abs(op).max() < 1e-13 and np.abs(ou).max() < 1e-12
    rng = np.random.default_rng(0)
    V, W = rng.uniform(-1, 1, (3, orc.n_u)), rng.uniform(-1, 1, (3, orc.n_u))
    kv, _ = orc.apply(V, np.zeros(orc.n_p)); kw, _ = orc.apply(W, np.zeros(orc.n_p))
    assert abs(np.vdot(W, kv) - np.vdot(V, kw)) < 1e-11 * abs(np.vdot(W, kv))
    # (B u, q) = -(u, B^T q) pairing: out_p . Q == -(out_u of pressure Q alone) . U
    Q = rng.uniform(-1, 1, orc.n_p)
    _, bu = orc.apply(V, np.zeros(orc.n_p))
    btq, _ = orc.apply(np.zeros((3, orc.n_u)), Q)
    assert abs(np.vdot(bu, Q) + np.vdot(btq, V)) < 1e-11 * abs(np.vdot(bu, Q))
    orc_d = oracle.StokesOracle(nc, verts, 63, 0.7)
    ou, _ = orc_d.apply(V, Q)
    nd = [2 * n + 1 for n in nc]
    o = ou.reshape(3, nd[2], nd[1], nd[0])
    assert np.abs(o[:, 0]).max() == 0 and np.abs(o[:, :, -1]).max() == 0 and np.abs(o[..., 0]).max() == 0
