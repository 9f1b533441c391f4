"""CPU oracle vs the independent numpy fixtures (tests/golden/*.npz, made by make_golden.py)
and the reference's own parity method (tests/tp_05dgp_support.cc:132-151: matrix-free apply of
every unit vector == assembled matrix column)."""
import glob
import os

import numpy as np
import pytest

TOL = 1e-12


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def load_cases(golden_dir):
    return sorted(glob.glob(os.path.join(golden_dir, "q*.npz")))


def make_oracle(o, g):
    orc = o.Oracle(int(g["p"]), g["ncell"], g["vertices"], int(g["mask"]))
    if "coef_lap" in g.files:
        orc.set_coefficient(1, g["coef_lap"])
    return orc


def test_fixtures_exist(golden_dir):
    assert len(load_cases(golden_dir)) == 7


@pytest.mark.parametrize("name", ["q1_cart_3x3x3", "q2_cart_2x2x2", "q2_pert_2x3x2",
                                  "q2_free_2x2x2", "q3_pert_2x2x2", "q4_cart_2x2x2",
                                  "q4_pert_3x2x2"])
def test_oracle_vs_numpy_fixture(name, golden_dir, oracle_mod):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    orc = make_oracle(oracle_mod, g)
    X = g["X"]
    for b in range(X.shape[0]):
        assert rel(orc.space_vmult(X[b], laplace=1.0), g["KX"][b]) < TOL
        assert rel(orc.space_vmult(X[b], mass=1.0), g["MX"][b]) < TOL
    assert rel(orc.st_vmult(g["Alpha"], g["Beta"], X), g["Y"]) < TOL
    assert rel(orc.st_vmult(g["Alpha"], g["Beta"], X, transpose=True), g["YT"]) < TOL
    assert rel(orc.diagonal(laplace=1.0), g["diagK"]) < TOL
    assert rel(orc.diagonal(mass=1.0), g["diagM"]) < TOL
    # vmult_slice (n x 1): first column of Alpha/Beta applied to one source block
    a1 = g["Alpha"][:, :1].copy()
    b1 = g["Beta"][:, :1].copy()
    ys = orc.st_vmult(a1, b1, X[:1])
    ref = a1 @ g["KX"][:1] + b1 @ g["MX"][:1]
    assert rel(ys, ref) < TOL
    # vmult_slice_add accumulates
    ys2 = orc.st_vmult(a1, b1, X[:1], dst=ys)
    assert rel(ys2, 2 * ref) < TOL


@pytest.mark.parametrize("name", ["q1_cart_3x3x3", "q2_cart_2x2x2", "q2_pert_2x3x2",
                                  "q2_free_2x2x2"])
def test_unit_vector_method(name, golden_dir, oracle_mod):
    """tp_05dgp_support.cc:140-149: || A_mf e_i - A_dense e_i ||_2 for every i."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    orc = make_oracle(oracle_mod, g)
    K = orc.dense(laplace=1.0)
    M = orc.dense(mass=1.0)
    scale = np.abs(g["K"]).max()
    assert np.abs(K - g["K"]).max() < 1e-13 * scale
    assert np.abs(M - g["M"]).max() < 1e-13 * np.abs(g["M"]).max()
    assert np.abs(K - K.T).max() < 1e-13 * scale
    assert np.abs(M - M.T).max() < 1e-14


def test_analytic_properties(oracle_mod):
    """K*1 = 0 and 1^T M 1 = |Omega| without constraints."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import structured_vertices
    v = structured_vertices((3, 2, 2), (0, 0, 0), (1.5, 1.0, 0.5), jitter=0.15, seed=3)
    for p in (1, 2, 3, 4):
        orc = oracle_mod.Oracle(p, (3, 2, 2), v, dirichlet_mask=0)
        one = np.ones(orc.n_dofs)
        assert np.abs(orc.space_vmult(one, laplace=1.0)).max() < 1e-12
        assert abs(one @ orc.space_vmult(one, mass=1.0) - 0.75) < 1e-13


def test_coefficient_replaces_scaling(oracle_mod):
    """operators.h:1152-1162: has_coefficient ? coefficient(cell,q) : scaling."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import structured_vertices
    v = structured_vertices((2, 2, 2), (-1, -1, -1), (1, 1, 1))
    orc = oracle_mod.Oracle(2, (2, 2, 2), v, 63)
    x = np.random.default_rng(0).uniform(-1, 1, orc.n_dofs)
    y1 = orc.space_vmult(x, laplace=1.0)
    orc.set_coefficient(1, np.full((orc.n_cells, 27), 3.0))
    y3 = orc.space_vmult(x, laplace=1.0)
    assert rel(y3, 3 * y1) < 1e-14
    y3b = orc.space_vmult(x, laplace=7.0)  # scaling value ignored once a coefficient is set
    assert rel(y3b, 3 * y1) < 1e-14


def test_coefficient_function(oracle_mod):
    """operators.h:883-891 regions on [-1,1]^3 and the per-coarse-cell table layout."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import structured_vertices
    nc = (10, 10, 5)
    v = structured_vertices(nc, (-1, -1, -1), (1, 1, 1))
    orc = oracle_mod.Oracle(1, nc, v, 63)
    c = orc.coefficient_values(1, 9, 16, 0.0, (5, 5, 5), (-1, -1, -1), (1, 1, 1))
    pts = orc.quadrature_points()
    exp = np.where(pts[..., 1] >= 0.2, np.where(pts[..., 0] < 0.2, 9.0, 16.0), 1.0)
    assert np.array_equal(c, exp)
    cd = orc.coefficient_values(1, 9, 16, 0.5, (5, 5, 5), (-1, -1, -1), (1, 1, 1))
    f = cd / c
    assert f.min() >= 0.5 and f.max() < 1.5
    # constant on each coarse (subdivision) cell: 125 distinct factors
    assert len(np.unique(np.round(f, 12))) == 125
    # first table entry (coarse cell 0,0,0) = first draw of mt19937(5489): 3499211612 / 2^32
    first = 0.5 + 3499211612 / 4294967296.0
    assert abs(f[0, 0] - first) < 1e-15


@pytest.mark.parametrize("name", ["stokes_nitsche_cart_2x2x2", "stokes_nitsche_pert_2x3x2", "stokes_nitsche_pert_3x2x2"])
def test_stokes_oracle_nitsche_faces_vs_dense_fixture(name, oracle_mod, golden_dir):
    """the boundary-face loop of the Stokes oracle (weak Nitsche faces, operators.h:1713-1741, and the functional of
    StokesNitscheMatrixFreeOperator, 1898-1940) against the independent dense numpy assembly"""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    o = oracle_mod.StokesOracle(tuple(g["ncell"]), g["vertices"], int(g["mask"]), float(g["nu"]), weak_mask=int(g["weak"]),
                                penalty1=float(g["penalty1"]), penalty2=float(g["penalty2"]))
    ou, op = o.apply(g["U"], g["P"])
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)  # noqa: E731
    assert rel(ou, g["SU"]) < 1e-13 and rel(op, g["SP"]) < 1e-13
    assert np.abs(o.face_points() - g["face_points"]).max() < 1e-14
    fu, fp = o.nitsche_rhs(g["G"])
    assert rel(fu, g["FU"]) < 1e-13 and rel(fp, g["FP"]) < 1e-13


@pytest.mark.parametrize("name", ["stokes_dgp_cart_2x2x2", "stokes_dgp_pert_2x3x2"])
def test_stokes_oracle_dg_pressure_vs_dense_fixture(name, oracle_mod, golden_dir):
    """FE_DGP(1) pressure (the reference's dGPressure) in the Stokes oracle - cell loop, Nitsche faces, Dirichlet functional -
    against the independent dense numpy assembly with deal.II's Legendre basis"""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    o = oracle_mod.StokesOracle(tuple(g["ncell"]), g["vertices"], int(g["mask"]), float(g["nu"]), weak_mask=int(g["weak"]), dg_pressure=True)
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)  # noqa: E731
    ou, op = o.apply(g["U"], g["P"])
    assert rel(ou, g["SU"]) < 1e-13 and rel(op, g["SP"]) < 1e-13
    mu, _ = o.apply(g["U"], g["P"], 0.0, 1.0)
    assert rel(mu, g["MU"]) < 1e-13
    if int(g["weak"]):
        fu, fp = o.nitsche_rhs(g["G"])
        assert rel(fu, g["FU"]) < 1e-13 and rel(fp, g["FP"]) < 1e-13
