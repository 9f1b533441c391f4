"""GPU parity tests proper: the HIP path, called through the C-ABI, against
 (a) the committed numpy golden fixtures and (b) the CPU oracle on seeded inputs.
Tolerance: rel-L2 <= 1e-12 (BASELINE.json north_star: 'within 1e-12 rel-L2 of the CPU reference')."""
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-12


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope="module")
def stfem():
    mod = importlib.import_module("dealii-stfem_amd")
    mod.lib()  # raises if the HIP library is missing: no fallback
    return mod


def apply(stfem, ctx, Alpha, Beta, X, transpose=False, add_to=None):
    A = stfem.SystemMatrix(ctx, Alpha, Beta)
    nsrc = Alpha.shape[0] if transpose else Alpha.shape[1]
    ndst = Alpha.shape[1] if transpose else Alpha.shape[0]
    src = stfem.BlockVector(ctx, nsrc).upload(X)
    dst = stfem.BlockVector(ctx, ndst)
    if add_to is not None:
        dst.upload(add_to)
        A._apply(dst, src, transpose, True, None)
    elif transpose:
        A.Tvmult(dst, src)
    else:
        A.vmult(dst, src)
    return dst.download()


CART_FIXTURES = ["q1_cart_3x3x3", "q2_cart_2x2x2", "q2_free_2x2x2", "q4_cart_2x2x2"]


@pytest.mark.parametrize("name", CART_FIXTURES)
def test_golden_fixture(name, stfem, golden_dir):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    ctx = stfem.MatrixFreeOperator(int(g["p"]), g["ncell"], vertices=g["vertices"],
                                   dirichlet_mask=int(g["mask"]))
    assert ctx.is_cartesian
    assert ctx.n_dofs == g["X"].shape[1]
    Y = apply(stfem, ctx, g["Alpha"], g["Beta"], g["X"])
    assert rel(Y, g["Y"]) < TOL
    YT = apply(stfem, ctx, g["Alpha"], g["Beta"], g["X"], transpose=True)
    assert rel(YT, g["YT"]) < TOL
    # MatrixFreeOperator::vmult, K = (0,1) and M = (1,0)
    nb = g["X"].shape[0]
    for ms, ls, ref in ((0.0, 1.0, g["KX"]), (1.0, 0.0, g["MX"])):
        op = stfem.MatrixFreeOperator(int(g["p"]), g["ncell"], vertices=g["vertices"],
                                      dirichlet_mask=int(g["mask"]), mass_matrix_scaling=ms,
                                      laplace_matrix_scaling=ls)
        src = stfem.BlockVector(op, 1).upload(g["X"][:1])
        dst = stfem.BlockVector(op, 1)
        op.vmult(dst, src)
        assert rel(dst.download(), ref[:1]) < TOL


GENERAL_FIXTURES = ["q2_pert_2x3x2", "q3_pert_2x2x2", "q4_pert_3x2x2"]


@pytest.mark.parametrize("name", GENERAL_FIXTURES)
def test_golden_fixture_general_mesh(name, stfem, golden_dir):
    """Perturbed (MappingQ1) cells with a per-quadrature-point laplace coefficient: the
    general-geometry kernel against the independent numpy assembly."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    ctx = stfem.MatrixFreeOperator(int(g["p"]), g["ncell"], vertices=g["vertices"],
                                   dirichlet_mask=int(g["mask"]))
    assert not ctx.is_cartesian
    ctx.evaluate_coefficient(g["coef_lap"], which=1)
    assert rel(apply(stfem, ctx, g["Alpha"], g["Beta"], g["X"]), g["Y"]) < TOL
    assert rel(apply(stfem, ctx, g["Alpha"], g["Beta"], g["X"], transpose=True), g["YT"]) < TOL
    op = stfem.MatrixFreeOperator(int(g["p"]), g["ncell"], vertices=g["vertices"],
                                  dirichlet_mask=int(g["mask"]), laplace_matrix_scaling=1.0)
    op.evaluate_coefficient(g["coef_lap"], which=1)
    dst = stfem.BlockVector(op, 1)
    op.vmult(dst, stfem.BlockVector(op, 1).upload(g["X"][:1]))
    assert rel(dst.download(), g["KX"][:1]) < TOL
    mop = stfem.MatrixFreeOperator(int(g["p"]), g["ncell"], vertices=g["vertices"],
                                   dirichlet_mask=int(g["mask"]), mass_matrix_scaling=1.0)
    mop.vmult(dst2 := stfem.BlockVector(mop, 1), stfem.BlockVector(mop, 1).upload(g["X"][:1]))
    assert rel(dst2.download(), g["MX"][:1]) < TOL


GENERAL_CASES = [
    # p, ncell, distort, mask, type, r, nsteps
    (4, (7, 6, 5), 0.15, 63, "CGP", 2, 1),   # cfg 2 shape (perturbed hypercube), small
    (2, (9, 5, 4), 0.15, 0, "CGP", 1, 4),
    (3, (5, 4, 6), 0.1, 63, "DG", 2, 1),
    (1, (4, 3, 3), 0.2, 0b101010, "DG", 1, 2),
]


@pytest.mark.parametrize("case", GENERAL_CASES, ids=lambda c: f"Q{c[0]}-{c[4]}{c[5]}x{c[6]}-pert")
def test_general_mesh_vs_oracle(case, stfem, oracle_mod):
    p, nc, distort, mask, tt, r, ns = case
    t = stfem.CGP if tt == "CGP" else stfem.DG
    Alpha, Beta, Gamma, Zeta = stfem.get_fe_time_weights(t, r, 0.02, ns)
    verts = stfem.mesh_vertices(nc, (0, 0, 0), (1, 1, 1), distort, 5489)
    ctx = stfem.MatrixFreeOperator(p, nc, vertices=verts, dirichlet_mask=mask)
    assert not ctx.is_cartesian
    orc = oracle_mod.Oracle(p, nc, verts, mask)
    X = random_blocks(Alpha.shape[0], ctx.n_dofs)
    assert rel(apply(stfem, ctx, Alpha, Beta, X), orc.st_vmult(Alpha, Beta, X)) < TOL
    assert rel(apply(stfem, ctx, Alpha, Beta, X, transpose=True),
               orc.st_vmult(Alpha, Beta, X, transpose=True)) < TOL
    g = Gamma if np.any(Gamma) else Zeta
    z = Zeta if np.any(Zeta) else Gamma
    ref = orc.st_vmult(g, z, X[:1])
    assert rel(apply(stfem, ctx, g, z, X[:1], add_to=ref), 2 * ref) < TOL
    # per-quadrature-point coefficients on both operators
    rng = np.random.default_rng(3)
    cl = rng.uniform(0.5, 2.0, (ctx.n_cells, (p + 1) ** 3)); cm = rng.uniform(0.5, 2.0, (ctx.n_cells, (p + 1) ** 3))
    ctx.evaluate_coefficient(cl, which=1); ctx.evaluate_coefficient(cm, which=0)
    orc.set_coefficient(1, cl); orc.set_coefficient(0, cm)
    assert rel(apply(stfem, ctx, Alpha, Beta, X), orc.st_vmult(Alpha, Beta, X)) < TOL


def test_per_q_coefficient_on_cartesian_mesh(stfem, oracle_mod):
    p, nc = 2, (5, 4, 3)
    verts = stfem.mesh_vertices(nc)
    ctx = stfem.MatrixFreeOperator(p, nc, vertices=verts)
    assert ctx.is_cartesian
    orc = oracle_mod.Oracle(p, nc, verts, 63)
    Alpha, Beta, _, _ = stfem.get_fe_time_weights(stfem.CGP, 2, 0.1, 1)
    X = random_blocks(2, ctx.n_dofs)
    cl = np.random.default_rng(8).uniform(0.5, 2.0, (ctx.n_cells, 27))
    ctx.evaluate_coefficient(cl, which=1); orc.set_coefficient(1, cl)
    assert rel(apply(stfem, ctx, Alpha, Beta, X), orc.st_vmult(Alpha, Beta, X)) < TOL
    ctx.evaluate_coefficient(None, which=1); orc.set_coefficient(1, None)
    assert rel(apply(stfem, ctx, Alpha, Beta, X), orc.st_vmult(Alpha, Beta, X)) < TOL


def random_blocks(n_blocks, n, seed=1234):
    return np.stack([np.random.default_rng(seed + b).uniform(-1, 1, n) for b in range(n_blocks)])


CASES = [
    # p, ncell, lower, upper, mask, time type, r, nsteps, tau
    (2, (16, 16, 16), (0, 0, 0), (1, 1, 1), 63, "CGP", 1, 4, 1 / 32),      # cfg 0 (tp_01 -d 3)
    (4, (6, 5, 4), (0, 0, 0), (1, 1, 1), 63, "CGP", 2, 1, 1 / 144),        # cfg 1 small
    (4, (7, 3, 2), (0, 0, 0), (2, 1, 0.5), 0, "CGP", 2, 2, 0.01),          # ragged, nb=4, no BC
    (3, (5, 4, 6), (-1, -1, -1), (1, 1, 1), 63, "DG", 2, 1, 1 / 64),       # cfg 3 shape
    (3, (3, 3, 3), (-1, -1, -1), (1, 1, 1), 0b010101, "DG", 1, 3, 0.1),    # nb=6
    (1, (9, 7, 5), (0, 0, 0), (1, 1, 1), 63, "DG", 0, 1, 0.2),             # nb=1, Q1
    (2, (4, 4, 3), (0, 0, 0), (1, 1, 1), 63, "CGP", 3, 4, 0.05),           # nb=12 -> tiled launches
    (1, (1, 1, 1), (0, 0, 0), (1, 1, 1), 0, "CGP", 1, 1, 1.0),             # single cell
    (4, (5, 9, 3), (0, 0, 0), (1, 1, 1), 63, "CGP", 4, 2, 0.02),           # Q4, nb=8: the largest instantiation
    (3, (9, 5, 7), (0, 0, 0), (1, 2, 1), 0b100110, "DG", 3, 2, 0.02),      # Q3, nb=8
    (4, (13, 5, 9), (0, 0, 0), (1, 1, 1), 63, "DG", 2, 1, 0.02),           # Q4, nb=3 (two cell groups per wave)
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"Q{c[0]}-{c[5]}{c[6]}x{c[7]}-{'x'.join(map(str, c[1]))}")
def test_vs_oracle(case, stfem, oracle_mod):
    p, nc, lo, up, mask, tt, r, ns, tau = case
    t = stfem.CGP if tt == "CGP" else stfem.DG
    Alpha, Beta, Gamma, Zeta = stfem.get_fe_time_weights(t, r, tau, ns)
    verts = stfem.mesh_vertices(nc, lo, up)
    ctx = stfem.MatrixFreeOperator(p, nc, vertices=verts, dirichlet_mask=mask)
    orc = oracle_mod.Oracle(p, nc, verts, mask)
    nb = Alpha.shape[0]
    X = random_blocks(nb, ctx.n_dofs)
    assert rel(apply(stfem, ctx, Alpha, Beta, X), orc.st_vmult(Alpha, Beta, X)) < TOL
    # every Cartesian system runs the pencil sweep since round 3 (4 - 8 blocks: PencilCore::middle_stream; Q4 x 8
    # blocks and systems of more than eight blocks in panels), unless STFEM_VARIANT asks for another kernel
    if not os.environ.get("STFEM_VARIANT"):
        assert ctx.last_kernel_name.startswith("st_sweep_pencil"), ctx.last_kernel_name
    assert rel(apply(stfem, ctx, Alpha, Beta, X, transpose=True),
               orc.st_vmult(Alpha, Beta, X, transpose=True)) < TOL
    # vmult_slice / vmult_slice_add (rhs assembly, operators.h:377-382, 586-611)
    g = Gamma if np.any(Gamma) else Zeta
    z = Zeta if np.any(Zeta) else Gamma
    ref = orc.st_vmult(g, z, X[:1])
    got = apply(stfem, ctx, g, z, X[:1])
    assert rel(got, ref) < TOL
    got2 = apply(stfem, ctx, g, z, X[:1], add_to=ref)
    assert rel(got2, 2 * ref) < TOL


def test_wave_matrices_and_coefficient(stfem, oracle_mod):
    """cfg 3 ingredients: Q3 x dG(2) wave matrices, c in {1,9,16} with per-coarse-cell factor."""
    p, nc = 3, (10, 10, 5)
    lo, up = (-1, -1, -1), (1, 1, 1)
    A, B, _, _, _ = stfem.get_fe_time_weights_wave(stfem.DG, 2, 1 / 64, 1)
    verts = stfem.mesh_vertices(nc, lo, up)
    coef = stfem.coefficient_per_cell(nc, verts, 1, 9, 16, 0.5, (5, 5, 5), lo, up)
    ctx = stfem.MatrixFreeOperator(p, nc, vertices=verts)
    ctx.evaluate_coefficient(coef, which=1)
    orc = oracle_mod.Oracle(p, nc, verts, 63)
    orc.set_coefficient(1, orc.coefficient_values(1, 9, 16, 0.5, (5, 5, 5), lo, up))
    X = random_blocks(3, ctx.n_dofs)
    assert rel(apply(stfem, ctx, A, B, X), orc.st_vmult(A, B, X)) < TOL
    # mass coefficient too
    cm = np.random.default_rng(5).uniform(0.5, 2.0, ctx.n_cells)
    ctx.evaluate_coefficient(cm, which=0)
    orc.set_coefficient(0, np.repeat(cm[:, None], (p + 1) ** 3, axis=1))
    assert rel(apply(stfem, ctx, A, B, X), orc.st_vmult(A, B, X)) < TOL
    ctx.evaluate_coefficient(None, which=0)
    ctx.evaluate_coefficient(None, which=1)
    orc.set_coefficient(0, None); orc.set_coefficient(1, None)
    assert rel(apply(stfem, ctx, A, B, X), orc.st_vmult(A, B, X)) < TOL


@pytest.mark.parametrize("p,nc,tt,r,ns", [(2, (11, 6, 5), "CGP", 1, 4), (4, (8, 5, 3), "CGP", 2, 2), (3, (7, 4, 5), "DG", 2, 2), (1, (9, 8, 7), "DG", 1, 4),
                                          (4, (5, 3, 3), "CGP", 4, 2)],
                         ids=["Q2-CGP1x4", "Q4-CGP2x2", "Q3-DG2x2", "Q1-DG1x4", "Q4-CGP4x2"])
def test_many_blocks_with_cell_coefficients(p, nc, tt, r, ns, stfem, oracle_mod):
    """4 / 6 / 8 temporal blocks (the reference's n_timesteps_at_once systems, fe_time.h:373-402) with cell-wise
    coefficients on K and M (operators.h:1060-1087): the streamed middle phase of the pencil sweep with its LDS weight table."""
    t = stfem.CGP if tt == "CGP" else stfem.DG
    Alpha, Beta, _, _ = stfem.get_fe_time_weights(t, r, 0.03, ns)
    verts = stfem.mesh_vertices(nc, (0, 0, 0), (1.0, 0.7, 1.3))
    ctx = stfem.MatrixFreeOperator(p, nc, vertices=verts, dirichlet_mask=0b011011)
    orc = oracle_mod.Oracle(p, nc, verts, 0b011011)
    rng = np.random.default_rng(17)
    cl, cm = rng.uniform(0.5, 3.0, ctx.n_cells), rng.uniform(0.5, 2.0, ctx.n_cells)
    ctx.evaluate_coefficient(cl, which=1); ctx.evaluate_coefficient(cm, which=0)
    orc.set_coefficient(1, np.repeat(cl, (p + 1) ** 3)); orc.set_coefficient(0, np.repeat(cm, (p + 1) ** 3))
    X = random_blocks(Alpha.shape[0], ctx.n_dofs)
    assert rel(apply(stfem, ctx, Alpha, Beta, X), orc.st_vmult(Alpha, Beta, X)) < TOL
    if not os.environ.get("STFEM_VARIANT"):
        assert ctx.last_kernel_name.startswith("st_sweep_pencil"), ctx.last_kernel_name
    assert rel(apply(stfem, ctx, Alpha, Beta, X, transpose=True), orc.st_vmult(Alpha, Beta, X, transpose=True)) < TOL
    ref = orc.st_vmult(Alpha, Beta, X)
    assert rel(apply(stfem, ctx, Alpha, Beta, X, add_to=ref), 2 * ref) < TOL  # dst += (compiler-tracked loads)


def test_rectangular_and_errors(stfem, oracle_mod):
    p, nc = 2, (3, 4, 2)
    verts = stfem.mesh_vertices(nc)
    ctx = stfem.MatrixFreeOperator(p, nc, vertices=verts)
    orc = oracle_mod.Oracle(p, nc, verts, 63)
    rng = np.random.default_rng(11)
    Alpha = rng.uniform(-1, 1, (3, 2)); Beta = rng.uniform(-1, 1, (3, 2))
    Alpha[1, 0] = 0.0
    X = random_blocks(2, ctx.n_dofs)
    assert rel(apply(stfem, ctx, Alpha, Beta, X), orc.st_vmult(Alpha, Beta, X)) < TOL
    X3 = random_blocks(3, ctx.n_dofs)
    assert rel(apply(stfem, ctx, Alpha, Beta, X3, transpose=True),
               orc.st_vmult(Alpha, Beta, X3, transpose=True)) < TOL
    A = stfem.SystemMatrix(ctx, Alpha, Beta)
    v2 = stfem.BlockVector(ctx, 2); v3 = stfem.BlockVector(ctx, 3)
    with pytest.raises(stfem.StfemError) as e:
        A.vmult(v2, v3)  # swapped block counts
    assert e.value.status == -5
    sq = stfem.SystemMatrix(ctx, np.eye(2), np.eye(2))
    with pytest.raises(stfem.StfemError) as e:
        sq.vmult(v2, v2)  # aliasing
    assert e.value.status == -6
    with pytest.raises(stfem.StfemError) as e:
        stfem.MatrixFreeOperator(7, nc)
    assert e.value.status == -2


def test_linearity_and_constrained_rows(stfem):
    """Size-independent properties at a larger size: linearity, Dirichlet rows stay zero,
    and K annihilates constants on an unconstrained mesh."""
    p, nc = 4, (12, 12, 12)
    Alpha, Beta, _, _ = stfem.get_fe_time_weights(stfem.CGP, 2, 1 / 144, 1)
    ctx = stfem.MatrixFreeOperator(p, nc)
    X = random_blocks(2, ctx.n_dofs); Y = random_blocks(2, ctx.n_dofs, seed=99)
    AX = apply(stfem, ctx, Alpha, Beta, X); AY = apply(stfem, ctx, Alpha, Beta, Y)
    AXY = apply(stfem, ctx, Alpha, Beta, 2.0 * X - 3.0 * Y)
    assert rel(AXY, 2.0 * AX - 3.0 * AY) < 1e-13
    n = p * nc[0] + 1
    grid = AX.reshape(2, n, n, n)
    for sl in (grid[:, 0], grid[:, -1], grid[:, :, 0], grid[:, :, -1], grid[:, :, :, 0], grid[:, :, :, -1]):
        assert np.all(sl == 0.0)
    free = stfem.MatrixFreeOperator(p, nc, dirichlet_mask=0, laplace_matrix_scaling=1.0)
    one = stfem.BlockVector(free, 1).upload(np.ones((1, free.n_dofs)))
    out = stfem.BlockVector(free, 1)
    free.vmult(out, one)
    assert np.abs(out.download()).max() < 1e-11
    mass = stfem.MatrixFreeOperator(p, nc, dirichlet_mask=0, mass_matrix_scaling=1.0)
    mass.vmult(out2 := stfem.BlockVector(mass, 1), stfem.BlockVector(mass, 1).upload(np.ones((1, mass.n_dofs))))
    assert abs(out2.download().sum() - 1.0) < 1e-12


def test_blas1_and_planes(stfem):
    p, nc = 2, (3, 3, 3)
    ctx = stfem.MatrixFreeOperator(p, nc)
    n = ctx.n_dofs
    X = random_blocks(3, n); Y = random_blocks(2, n, seed=7)
    A = np.array([[1.5, 0.0, -2.0], [0.0, 0.25, 1.0]])
    b = stfem.BlockVector(ctx, 3).upload(X); c = stfem.BlockVector(ctx, 2).upload(Y)
    stfem.tensorproduct_add(ctx, c, A, b)
    assert rel(c.download(), Y + A @ X) < 1e-15
    assert abs(stfem.dot(ctx, b, b) - np.sum(X * X)) < 1e-10
    plane = 7 * 7
    assert abs(stfem.dot(ctx, b, b, n_own=n - plane) - np.sum(X[:, :n - plane] ** 2)) < 1e-10


@pytest.mark.parametrize("number", ["double", "float"])
def test_gram_schmidt_step_on_the_device(number, stfem):
    """stfem_multi_dot / stfem_multi_axpy / stfem_orthogonalize (the Gram-Schmidt step of SolverFGMRES / SolverGMRES): against
    numpy, and bitwise reproducible (two-stage reductions with a fixed order; stfem_dot too)"""
    ctx = stfem.MatrixFreeOperator(2, (9, 7, 5), number=number)
    nb, k = 3, 11
    rng = np.random.default_rng(2)
    f = (lambda a: a.astype(np.float32).astype(np.float64)) if number == "float" else (lambda a: a)
    V = [f(rng.uniform(-1, 1, (nb, ctx.n_dofs))) for _ in range(k)]
    W = f(rng.uniform(-1, 1, (nb, ctx.n_dofs)))
    vs = [stfem.BlockVector(ctx, nb).upload(v) for v in V]
    w = stfem.BlockVector(ctx, nb).upload(W)
    tol = 1e-13 if number == "double" else 1e-6
    want = np.array([np.sum(v * W) for v in V])
    got = stfem.multi_dot(ctx, vs, w)
    assert np.allclose(got, want, rtol=tol, atol=tol * np.abs(want).max())
    assert np.array_equal(got, stfem.multi_dot(ctx, vs, w))
    d1 = stfem.dot(ctx, vs[0], w)
    assert d1 == stfem.dot(ctx, vs[0], w) and d1 == got[0]
    coef = rng.uniform(-1, 1, k)
    stfem.multi_axpy(ctx, coef, vs, w)
    W2 = w.download()
    assert rel(W2, W + sum(c * v for c, v in zip(coef, V))) < (1e-14 if number == "double" else 1e-6)
    h, n2, b2 = stfem.orthogonalize(ctx, vs, w)
    assert abs(b2 - np.sum(W2 ** 2)) <= (1e-12 if number == "double" else 1e-5) * b2
    hw = np.array([np.sum(v * W2) for v in V])
    assert np.allclose(h, hw, rtol=tol, atol=tol * np.abs(hw).max())
    W3 = W2 - sum(c * v for c, v in zip(h, V))
    assert rel(w.download(), W3) < (1e-13 if number == "double" else 2e-6)
    assert abs(n2 - np.sum(w.download() ** 2)) <= (1e-12 if number == "double" else 1e-5) * n2


@pytest.mark.parametrize("name", CART_FIXTURES + GENERAL_FIXTURES)
def test_diagonal(name, stfem, golden_dir):
    """compute_diagonal (operators.h:1092-1110): forward diagonal of K and of M."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    for ms, ls, ref in ((0.0, 1.0, g["diagK"]), (1.0, 0.0, g["diagM"])):
        op = stfem.MatrixFreeOperator(int(g["p"]), g["ncell"], vertices=g["vertices"],
                                      dirichlet_mask=int(g["mask"]), mass_matrix_scaling=ms,
                                      laplace_matrix_scaling=ls)
        if "coef_lap" in g.files and ls != 0.0:
            op.evaluate_coefficient(g["coef_lap"], which=1)
        d = op.compute_diagonal().download()[0]
        assert rel(d, ref) < TOL


def test_diagonal_with_cell_coefficient_vs_oracle(stfem, oracle_mod):
    p, nc = 3, (4, 3, 2)
    verts = stfem.mesh_vertices(nc, (-1, -1, -1), (1, 1, 1))
    coef = stfem.coefficient_per_cell(nc, verts, 1, 9, 16, 0.5, (2, 1, 1), (-1, -1, -1), (1, 1, 1))
    op = stfem.MatrixFreeOperator(p, nc, vertices=verts, laplace_matrix_scaling=1.0, mass_matrix_scaling=0.5)
    op.evaluate_coefficient(coef, which=1)
    orc = oracle_mod.Oracle(p, nc, verts, 63)
    orc.set_coefficient(1, np.repeat(coef[:, None], (p + 1) ** 3, axis=1))
    assert rel(op.compute_diagonal().download()[0], orc.diagonal(mass=0.5, laplace=1.0)) < TOL


# ----------------------------------------------------------------------------- fp32 instantiation
# The reference instantiates the path for float as well (include/operators.cc:5-45) and runs the
# whole multigrid preconditioner in it (tests/tp_01.cc:780, 801-806).  Tolerance: 1e-5 rel-L2
# against the fp64 oracle.
TOL32 = 1e-5


def apply32(stfem, ctx, Alpha, Beta, X, transpose=False):
    A = stfem.SystemMatrix(ctx, Alpha, Beta)
    nsrc = Alpha.shape[0] if transpose else Alpha.shape[1]
    ndst = Alpha.shape[1] if transpose else Alpha.shape[0]
    src = stfem.BlockVector(ctx, nsrc).upload(X)
    dst = stfem.BlockVector(ctx, ndst)
    (A.Tvmult if transpose else A.vmult)(dst, src)
    return dst.download()


@pytest.mark.parametrize("name", CART_FIXTURES + GENERAL_FIXTURES)
def test_fp32_golden_fixture(name, stfem, golden_dir):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    ctx = stfem.MatrixFreeOperator(int(g["p"]), g["ncell"], vertices=g["vertices"],
                                   dirichlet_mask=int(g["mask"]), number="float")
    assert stfem.lib().stfem_ctx_precision(ctx._h) == 1
    if "coef_lap" in g.files:
        ctx.evaluate_coefficient(g["coef_lap"], which=1)
    assert rel(apply32(stfem, ctx, g["Alpha"], g["Beta"], g["X"]), g["Y"]) < TOL32
    assert rel(apply32(stfem, ctx, g["Alpha"], g["Beta"], g["X"], transpose=True), g["YT"]) < TOL32
    d = stfem.MatrixFreeOperator(int(g["p"]), g["ncell"], vertices=g["vertices"], dirichlet_mask=int(g["mask"]),
                                 mass_matrix_scaling=1.0, number="float").compute_diagonal().download()[0]
    assert rel(d, g["diagM"]) < TOL32


@pytest.mark.parametrize("case", [CASES[0], CASES[1], CASES[3], CASES[6], CASES[8], CASES[9]],
                         ids=lambda c: f"f32-Q{c[0]}-{c[5]}{c[6]}x{c[7]}")
def test_fp32_vs_oracle(case, stfem, oracle_mod):
    p, nc, lo, up, mask, tt, r, ns, tau = case
    t = stfem.CGP if tt == "CGP" else stfem.DG
    Alpha, Beta, Gamma, Zeta = stfem.get_fe_time_weights(t, r, tau, ns)
    verts = stfem.mesh_vertices(nc, lo, up)
    ctx = stfem.MatrixFreeOperator(p, nc, vertices=verts, dirichlet_mask=mask, number="float")
    orc = oracle_mod.Oracle(p, nc, verts, mask)
    X = random_blocks(Alpha.shape[0], ctx.n_dofs)
    assert rel(apply32(stfem, ctx, Alpha, Beta, X), orc.st_vmult(Alpha, Beta, X)) < TOL32
    assert rel(apply32(stfem, ctx, Alpha, Beta, X, transpose=True),
               orc.st_vmult(Alpha, Beta, X, transpose=True)) < TOL32
    # BLAS-1 in the context's precision, dot accumulated in double
    a = stfem.BlockVector(ctx, 1).upload(X[:1]); b = stfem.BlockVector(ctx, 1).upload(X[:1])
    stfem.tensorproduct_add(ctx, a, np.array([[0.5]]), b)
    assert rel(a.download(), 1.5 * X[:1]) < 1e-6
    assert abs(stfem.dot(ctx, b, b) - np.sum(X[:1].astype(np.float32).astype(np.float64) ** 2)) < 1e-6 * np.sum(X[:1] ** 2)


def _random_case(seed):
    rng = np.random.default_rng(1000 + seed)
    p = int(rng.integers(1, 5))
    nc = tuple(int(v) for v in rng.integers(1, 16 if p <= 2 else 14, size=3))
    tt = "CGP" if rng.random() < 0.6 else "DG"
    r = int(rng.integers(1, 4)) if tt == "CGP" else int(rng.integers(0, 3))
    ns = int(rng.integers(1, 3))
    mask = int(rng.integers(0, 64))
    up = tuple(float(v) for v in rng.uniform(0.5, 2.0, size=3))
    pert = bool(rng.random() < 0.3)
    coef = bool(rng.random() < 0.4)
    return p, nc, up, mask, tt, r, ns, pert, coef


@pytest.mark.parametrize("seed", range(24))
def test_random_configurations_vs_oracle(seed, stfem, oracle_mod):
    """Seeded random sweep over degree, ragged mesh sizes (tiles, chunks and cell groups that do not
    divide the mesh), block counts, Dirichlet masks, anisotropic boxes, perturbed meshes and per-cell
    coefficients: vmult and Tvmult against the oracle."""
    p, nc, up, mask, tt, r, ns, pert, coef = _random_case(seed)
    t = stfem.CGP if tt == "CGP" else stfem.DG
    Alpha, Beta, _, _ = stfem.get_fe_time_weights(t, r, 0.05, ns)
    verts = stfem.mesh_vertices(nc, (0, 0, 0), up, distort=0.12 if pert else 0.0, seed=seed)
    ctx = stfem.MatrixFreeOperator(p, nc, vertices=verts, dirichlet_mask=mask)
    orc = oracle_mod.Oracle(p, nc, verts, mask)
    if coef:
        c = np.random.default_rng(seed).uniform(0.5, 3.0, ctx.n_cells)
        ctx.evaluate_coefficient(c, which=1)
        orc.set_coefficient(1, np.repeat(c, (p + 1) ** 3))
    nb = Alpha.shape[0]
    X = random_blocks(nb, ctx.n_dofs, seed=seed)
    assert rel(apply(stfem, ctx, Alpha, Beta, X), orc.st_vmult(Alpha, Beta, X)) < TOL
    assert rel(apply(stfem, ctx, Alpha, Beta, X, transpose=True),
               orc.st_vmult(Alpha, Beta, X, transpose=True)) < TOL
