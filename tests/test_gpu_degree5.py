"""FE_Q(5) in space (the reference's run-time degree: tests/tp_01.cc:76-78 builds FE_Q(fe_degree + 1), its golden tests/tp_01.output
holds the k = 4 tables of FE_Q(5) x cG(4)): the operator apply, the diagonal and the space transfers of degree-5 contexts against the
oracle.  Degree 5 runs the tile sweep on every mesh (the pencil sweep's planes do not fit a wave's registers); the cell-patch
smoother takes degree-5 blocks of one or two temporal blocks (512 rows per cell block at most)."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL, TOL32 = 1e-12, 1e-5


@pytest.fixture(scope="module")
def stfem():
    mod = importlib.import_module("dealii-stfem_amd")
    mod.lib()
    return mod


@pytest.fixture(scope="module")
def oracle_mod():
    from oracle import oracle
    return oracle


def rel(a, b):
    return np.linalg.norm(np.ravel(a) - np.ravel(b)) / max(np.linalg.norm(np.ravel(b)), 1e-300)


def blocks(nb, n, seed=77):
    return np.stack([np.random.default_rng(seed + b).uniform(-1, 1, n) for b in range(nb)])


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


CASES = [
    # ncell, upper, mask, time type, r, steps, distort
    ((4, 3, 2), (1, 1, 1), 63, "CGP", 2, 1, 0.0),         # two blocks
    ((5, 2, 3), (2, 1, 0.5), 0b100110, "DG", 1, 2, 0.0),  # four blocks, anisotropic cells, mixed boundary
    ((3, 3, 3), (1, 1, 1), 0, "DG", 0, 1, 0.0),           # one block, no constraints
    ((11, 2, 2), (1, 1, 1), 63, "CGP", 4, 2, 0.0),        # eight blocks (tests/tp_01.output k = 4: cG(4), two steps at once)
    ((2, 2, 5), (1, 1, 1), 63, "CGP", 3, 4, 0.0),         # twelve blocks: panels
    ((1, 1, 1), (1, 2, 3), 63, "CGP", 1, 1, 0.0),         # a single cell
    ((4, 3, 3), (1, 1, 1), 63, "CGP", 2, 1, 0.15),        # perturbed (MappingQ1) cells: stored metric
    ((3, 2, 4), (1, 1, 1), 0b010101, "DG", 1, 2, 0.1),    # perturbed, four blocks
]


@pytest.mark.parametrize("number", ["double", "float"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"Q5-{c[3]}{c[4]}x{c[5]}-{'x'.join(map(str, c[0]))}{'-pert' if c[6] else ''}")
def test_degree5_vs_oracle(case, number, stfem, oracle_mod):
    nc, up, mask, tt, r, ns, distort = case
    p, tol = 5, (TOL if number == "double" else TOL32)
    Alpha, Beta, Gamma, Zeta = stfem.get_fe_time_weights(stfem.CGP if tt == "CGP" else stfem.DG, r, 0.03, ns)
    verts = stfem.mesh_vertices(nc, (0, 0, 0), up, distort, 5489) if distort else stfem.mesh_vertices(nc, (0, 0, 0), up)
    ctx = stfem.MatrixFreeOperator(p, nc, vertices=verts, dirichlet_mask=mask, number=number)
    assert ctx.is_cartesian == (distort == 0.0) and ctx.n_dofs == int(np.prod([5 * n + 1 for n in nc]))
    orc = oracle_mod.Oracle(p, nc, verts, mask)
    X = blocks(Alpha.shape[0], ctx.n_dofs)
    assert rel(apply(stfem, ctx, Alpha, Beta, X), orc.st_vmult(Alpha, Beta, X)) < tol
    assert ctx.last_kernel_name.startswith("st_sweep_cart_tile"), ctx.last_kernel_name
    assert rel(apply(stfem, ctx, Alpha, Beta, X, transpose=True), orc.st_vmult(Alpha, Beta, X, transpose=True)) < tol
    g = Gamma if np.any(Gamma) else Zeta
    z = Zeta if np.any(Zeta) else Gamma
    ref = orc.st_vmult(g, z, X[:1])
    assert rel(apply(stfem, ctx, g, z, X[:1]), ref) < tol                     # vmult_slice
    assert rel(apply(stfem, ctx, g, z, X[:1], add_to=ref), 2 * ref) < tol     # vmult_slice_add
    # coefficients: one value per cell on axis-aligned cells, one per quadrature point on perturbed ones
    rng = np.random.default_rng(3)
    if distort:
        cl = rng.uniform(0.5, 2.0, (ctx.n_cells, 216))
        ctx.evaluate_coefficient(cl, which=1)
        orc.set_coefficient(1, cl)
    else:
        cc = rng.uniform(0.5, 2.0, ctx.n_cells)
        ctx.evaluate_coefficient(cc, which=1)
        orc.set_coefficient(1, np.repeat(cc[:, None], 216, axis=1))
    assert rel(apply(stfem, ctx, Alpha, Beta, X), orc.st_vmult(Alpha, Beta, X)) < tol


@pytest.mark.parametrize("distort", [0.0, 0.12])
def test_degree5_space_operators_and_diagonal(distort, stfem, oracle_mod):
    """MatrixFreeOperator::vmult with K = (0, 1), M = (1, 0) and compute_diagonal (operators.h:1019-1044, 1092-1110)"""
    p, nc, mask = 5, (3, 2, 3), 63
    verts = stfem.mesh_vertices(nc, (0, 0, 0), (1, 1.5, 1), distort, 11) if distort else stfem.mesh_vertices(nc, (0, 0, 0), (1, 1.5, 1))
    orc = oracle_mod.Oracle(p, nc, verts, mask)
    X = blocks(1, orc.n_dofs if hasattr(orc, "n_dofs") else int(np.prod([5 * n + 1 for n in nc])))
    for ms, ls in ((0.0, 1.0), (1.0, 0.0), (0.5, 2.0)):
        op = stfem.MatrixFreeOperator(p, nc, vertices=verts, dirichlet_mask=mask, mass_matrix_scaling=ms, laplace_matrix_scaling=ls)
        dst = stfem.BlockVector(op, 1)
        op.vmult(dst, stfem.BlockVector(op, 1).upload(X))
        assert rel(dst.download(), orc.st_vmult(np.array([[ls]]), np.array([[ms]]), X)) < TOL
        assert rel(op.compute_diagonal().download()[0], orc.diagonal(mass=ms, laplace=ls)) < TOL


@pytest.mark.parametrize("pf,ncf,pc,ncc", [(5, (2, 2, 2), 4, (2, 2, 2)), (5, (4, 2, 2), 5, (2, 1, 1)), (5, (2, 2, 4), 2, (1, 1, 2))])
def test_degree5_space_transfers(pf, ncf, pc, ncc, stfem):
    """MGTwoLevelTransfer between a FE_Q(5) level and a coarser one (p, h and hp): the table-driven 1D passes"""
    from oracle import stmg_oracle
    mask = 63
    fine, coarse = stfem.MatrixFreeOperator(pf, ncf, dirichlet_mask=mask), stfem.MatrixFreeOperator(pc, ncc, dirichlet_mask=mask)
    T = stfem.MGTwoLevelTransfer(fine, coarse)
    P = stmg_oracle.space_prolongation(pf, ncf, mask, pc, ncc, mask)
    rng = np.random.default_rng(4)
    Uc, Uf = rng.uniform(-1, 1, (2, coarse.n_dofs)), rng.uniform(-1, 1, (2, fine.n_dofs))
    uc, uf = stfem.BlockVector(coarse, 2).upload(Uc), stfem.BlockVector(fine, 2).upload(Uf)
    out_f, out_c = stfem.BlockVector(fine, 2), stfem.BlockVector(coarse, 2).upload(Uc)
    T.prolongate(out_f, uc)
    assert rel(out_f.download(), (P @ Uc.T).T) < 1e-13
    T.restrict_and_add(out_c, uf)
    assert rel(out_c.download(), Uc + (P.T @ Uf.T).T) < 1e-13


@pytest.mark.parametrize("number", ["double", "float"])
@pytest.mark.parametrize("nc,ttype,r,mask,distort", [((3, 2, 2), 0, 1, 63, 0.0), ((2, 2, 3), 0, 2, 63 & ~48, 0.0), ((2, 2, 2), 1, 0, 63, 0.1)])
def test_degree5_vanka_vs_oracle(nc, ttype, r, mask, distort, number, stfem):
    """PreconditionVanka on FE_Q(5) cells (216 rows per temporal block: one and two blocks fit the apply kernel), block classes on
    axis-aligned meshes and one block per cell on perturbed ones, against the dense restatement"""
    from oracle import vanka_oracle
    Alpha, Beta, _, _ = stfem.get_fe_time_weights(ttype, r, 0.05, 1)
    nb = Alpha.shape[0]
    verts = stfem.mesh_vertices(nc, distort=distort, seed=11) if distort else stfem.mesh_vertices(nc)
    ctx = (stfem.MatrixFreeOperator(5, nc, vertices=verts, number=number, dirichlet_mask=mask) if distort else
           stfem.MatrixFreeOperator(5, nc, number=number, dirichlet_mask=mask))
    V = stfem.PreconditionVanka(ctx, Alpha, Beta)
    ref = vanka_oracle.VankaOracle(5, nc, verts, mask, Alpha, Beta)
    rng = np.random.default_rng(7)
    X = rng.uniform(-1, 1, (nb, ctx.n_dofs))
    if number == "float":
        X = X.astype(np.float32).astype(np.float64)
    src, dst = stfem.BlockVector(ctx, nb).upload(X), stfem.BlockVector(ctx, nb).upload(rng.uniform(-1, 1, (nb, ctx.n_dofs)))
    V.vmult(dst, src)
    Y = dst.download()
    assert rel(Y, ref.vmult(X)) < (1e-10 if number == "double" else 5e-4)
    V.vmult(dst, src)
    assert np.array_equal(dst.download(), Y)


def test_degree5_limits(stfem):
    """what has no FE_Q(5) instantiation fails with a status, not with a wrong result: cell blocks of more than 512 rows (three
    temporal blocks of 216), degree 6"""
    ctx = stfem.MatrixFreeOperator(5, (2, 2, 2))
    Alpha, Beta, _, _ = stfem.get_fe_time_weights(stfem.CGP, 3, 0.1, 1)
    with pytest.raises(stfem.StfemError):
        stfem.PreconditionVanka(ctx, Alpha, Beta)
    with pytest.raises(stfem.StfemError):
        stfem.MatrixFreeOperator(6, (2, 2, 2))
