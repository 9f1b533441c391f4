"""Cell-patch Vanka smoother: the HIP apply (stfem_vanka_vmult: MFMA GEMM per block class with
fused gather / scatter, blocks built from Kronecker products of restricted 1D matrices) against the dense
numpy restatement of the reference's PreconditionVanka (oracle/vanka_oracle.py, stmg.h:619-907)."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b)


# the apply has two schemes: eight colour launches with the scatter fused in (large meshes) and, on meshes of up to 50 000 cells,
# all cells in one launch + one collecting launch (csrc/stfem_vanka.hip); STFEM_VANKA_COLOURS=1 (read when the smoother is
# created) forces the first on small meshes too
SCHEMES = ["two-phase", "colours"]


def _scheme(monkeypatch, scheme):
    if scheme == "colours":
        monkeypatch.setenv("STFEM_VANKA_COLOURS", "1")
    else:
        monkeypatch.delenv("STFEM_VANKA_COLOURS", raising=False)


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("number", ["double", "float"])
@pytest.mark.parametrize("p,nc,ttype,r,mask,upper,nsteps", [
    (2, (3, 3, 3), 0, 2, 63, (1.0, 1.0, 1.0), 1),
    (1, (5, 4, 3), 1, 0, 63, (1.0, 2.0, 0.5), 1),   # one temporal block, anisotropic cells
    (3, (4, 2, 3), 1, 1, 63 & ~48, (1.0, 1.0, 1.0), 1),  # dG(1), open z faces
    (4, (3, 2, 2), 0, 2, 63, (1.0, 1.0, 1.0), 1),   # cfg 1's element: 250 x 250 blocks
    (4, (2, 2, 2), 0, 2, 63, (1.0, 1.0, 1.0), 2),  # cG(2) with two time steps per slab on Q4: 500 x 500 blocks
    (2, (1, 1, 1), 0, 2, 63, (1.0, 1.0, 1.0), 1),   # a single cell: the exact inverse
    (2, (2, 1, 4), 0, 3, 63 & ~3, (1.0, 1.0, 1.0), 1),  # three temporal blocks, 81 rows -> padded tiles
])
def test_vanka_vs_oracle(p, nc, ttype, r, mask, upper, nsteps, number, scheme, monkeypatch):
    from oracle import vanka_oracle
    stfem = importlib.import_module("dealii-stfem_amd")
    _scheme(monkeypatch, scheme)
    Alpha, Beta, _, _ = stfem.get_fe_time_weights(ttype, r, 0.05, nsteps)
    nb = Alpha.shape[0]
    ctx = stfem.MatrixFreeOperator(p, nc, lower=(0, 0, 0), upper=upper, number=number, dirichlet_mask=mask)
    V = stfem.PreconditionVanka(ctx, Alpha, Beta)
    ref = vanka_oracle.VankaOracle(p, nc, stfem.mesh_vertices(nc, (0, 0, 0), upper), mask, Alpha, Beta)
    ncls = 1
    for d in range(3):
        ncls *= min(nc[d], 3)
    assert V.n_classes == ncls
    rng = np.random.default_rng(7)
    X = rng.uniform(-1, 1, (nb, ctx.n_dofs))
    if number == "float":
        X = X.astype(np.float32).astype(np.float64)
    src = stfem.BlockVector(ctx, nb).upload(X)
    dst = stfem.BlockVector(ctx, nb).upload(rng.uniform(-1, 1, (nb, ctx.n_dofs)))  # overwritten
    V.vmult(dst, src)
    Y = dst.download()
    want = ref.vmult(X)
    assert rel(Y, want) < (1e-10 if number == "double" else 2e-4)
    V.vmult(dst, src)  # deterministic: no atomics
    assert np.array_equal(dst.download(), Y)
    with pytest.raises(stfem.StfemError):
        V.vmult(src, src)


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("number", ["double", "float"])
@pytest.mark.parametrize("p,nc,ttype,r,mask,distort,coef", [
    (2, (3, 3, 2), 0, 2, 63, 0.12, None),        # perturbed mesh
    (4, (2, 2, 2), 0, 2, 63, 0.1, None),         # 250 x 250 blocks, one per cell
    (3, (3, 2, 3), 1, 1, 63 & ~12, 0.0, "cell"),  # Cartesian mesh, discontinuous laplace coefficient (cfg 3's setting)
    (2, (2, 3, 2), 0, 1, 63, 0.1, "q"),          # per-quadrature-point coefficient on a perturbed mesh
])
def test_vanka_per_cell_blocks_vs_oracle(p, nc, ttype, r, mask, distort, coef, number, scheme, monkeypatch):
    """General meshes and coefficient tables: one block per cell (set-up from device-computed cell matrices,
    HBM-streaming apply), against the same dense restatement."""
    from oracle import vanka_oracle
    stfem = importlib.import_module("dealii-stfem_amd")
    _scheme(monkeypatch, scheme)
    Alpha, Beta, _, _ = stfem.get_fe_time_weights(ttype, r, 0.05, 1)
    nb = Alpha.shape[0]
    verts = stfem.mesh_vertices(nc, distort=distort, seed=11)
    ctx = (stfem.MatrixFreeOperator(p, nc, vertices=verts, number=number, dirichlet_mask=mask) if distort else
           stfem.MatrixFreeOperator(p, nc, number=number, dirichlet_mask=mask))
    ncells, nq = nc[0] * nc[1] * nc[2], (p + 1) ** 3
    rng = np.random.default_rng(2)
    cl = None
    if coef == "cell":
        cc = rng.choice([1.0, 9.0, 16.0], ncells)
        ctx.evaluate_coefficient(cc, which=1)
        cl = np.repeat(cc, nq)  # the oracle takes one value per (cell, quadrature point)
    elif coef == "q":
        cl = rng.uniform(0.5, 2.0, ncells * nq)
        ctx.evaluate_coefficient(cl.reshape(ncells, nq), which=1)
    if number == "float" and cl is not None:
        cl = cl.astype(np.float32).astype(np.float64)
    V = stfem.PreconditionVanka(ctx, Alpha, Beta)
    assert V.n_classes == ncells
    ref = vanka_oracle.VankaOracle(p, nc, verts, mask, Alpha, Beta, coef_lap=cl)
    X = rng.uniform(-1, 1, (nb, ctx.n_dofs))
    if number == "float":
        X = X.astype(np.float32).astype(np.float64)
    src = stfem.BlockVector(ctx, nb).upload(X)
    dst = stfem.BlockVector(ctx, nb).upload(rng.uniform(-1, 1, (nb, ctx.n_dofs)))  # overwritten
    V.vmult(dst, src)
    Y = dst.download()
    assert rel(Y, ref.vmult(X)) < (1e-10 if number == "double" else 5e-4)
    V.vmult(dst, src)
    assert np.array_equal(dst.download(), Y)


@pytest.mark.parametrize("scheme", SCHEMES)
def test_vanka_blocks_at_descending_addresses(scheme, monkeypatch):
    """The block arrays of a vector may lie anywhere: the kernel's offset tables hold differences to block 0 that are
    negative when a later block sits at a lower address (found by the slab driver: FGMRES stagnated on 12^3 cells)."""
    from oracle import vanka_oracle
    stfem = importlib.import_module("dealii-stfem_amd")
    _scheme(monkeypatch, scheme)
    p, nc = 2, (3, 2, 3)
    Alpha, Beta, _, _ = stfem.get_fe_time_weights(stfem.CGP, 2, 0.05, 1)
    nb = Alpha.shape[0]
    ctx = stfem.MatrixFreeOperator(p, nc)
    V = stfem.PreconditionVanka(ctx, Alpha, Beta)
    ref = vanka_oracle.VankaOracle(p, nc, stfem.mesh_vertices(nc), 63, Alpha, Beta)
    rng = np.random.default_rng(4)
    X = rng.uniform(-1, 1, (nb, ctx.n_dofs))
    pool = stfem.BlockVector(ctx, 4)
    ptrs = sorted(pool.block_ptr(b) for b in range(4))
    for order in ((1, 0, 3, 2), (0, 1, 2, 3), (3, 2, 1, 0)):
        src = stfem.BlockVector(ctx, device_ptrs=[ptrs[order[0]], ptrs[order[1]]])
        dst = stfem.BlockVector(ctx, device_ptrs=[ptrs[order[2]], ptrs[order[3]]])
        src.upload(X)
        dst.upload(np.full((nb, ctx.n_dofs), 1e30))  # every entry must be overwritten
        V.vmult(dst, src)
        assert rel(dst.download(), ref.vmult(X)) < 1e-10, order


@pytest.mark.parametrize("number", ["double", "float"])
def test_vanka_device_setup_equals_host_setup(number, monkeypatch):
    """the per-cell blocks assembled and inverted on the device (batched Gauss-Jordan) against the same steps on the host"""
    stfem = importlib.import_module("dealii-stfem_amd")
    p, nc = 3, (3, 2, 4)
    Alpha, Beta, _, _ = stfem.get_fe_time_weights(0, 2, 0.05, 1)
    nb = Alpha.shape[0]
    ctx = stfem.MatrixFreeOperator(p, nc, vertices=stfem.mesh_vertices(nc, distort=0.12, seed=4), number=number, dirichlet_mask=63 & ~16)
    X = np.random.default_rng(2).uniform(-1, 1, (nb, ctx.n_dofs))
    out = []
    for host in ("0", "1"):
        monkeypatch.setenv("STFEM_VANKA_HOST_SETUP", host)
        V = stfem.PreconditionVanka(ctx, Alpha, Beta)
        assert V.n_classes == nc[0] * nc[1] * nc[2]
        src, dst = stfem.BlockVector(ctx, nb).upload(X), stfem.BlockVector(ctx, nb)
        V.vmult(dst, src)
        out.append(dst.download())
    assert rel(out[0], out[1]) < (1e-11 if number == "double" else 1e-5)


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("number", ["double", "float"])
@pytest.mark.parametrize("distort", [0.0, 0.1])
def test_vanka_relaxation_step(number, distort, scheme, monkeypatch):
    """stfem_vanka_step: dst = (dst | 0) + omega * V src, the step of PreconditionRelaxation (stmg.h:1199-1238) fused into the
    scatter, against vmult + a separate update; class blocks and one block per cell, both apply schemes."""
    stfem = importlib.import_module("dealii-stfem_amd")
    _scheme(monkeypatch, scheme)
    p, nc = 2, (4, 3, 3)
    Alpha, Beta, _, _ = stfem.get_fe_time_weights(stfem.CGP, 2, 0.05, 1)
    nb = Alpha.shape[0]
    ctx = (stfem.MatrixFreeOperator(p, nc, vertices=stfem.mesh_vertices(nc, distort=distort, seed=5), number=number) if distort else
           stfem.MatrixFreeOperator(p, nc, number=number))
    V = stfem.PreconditionVanka(ctx, Alpha, Beta)
    rng = np.random.default_rng(9)
    f = (lambda a: a.astype(np.float32).astype(np.float64)) if number == "float" else (lambda a: a)
    X, D0 = f(rng.uniform(-1, 1, (nb, ctx.n_dofs))), f(rng.uniform(-1, 1, (nb, ctx.n_dofs)))
    src = stfem.BlockVector(ctx, nb).upload(X)
    ref = stfem.BlockVector(ctx, nb)
    V.vmult(ref, src)
    Y = ref.download()
    tol = 1e-13 if number == "double" else 2e-6
    dst = stfem.BlockVector(ctx, nb).upload(D0)
    V.step(dst, 0.7, False, src)
    assert rel(dst.download(), 0.7 * Y) < tol
    dst.upload(D0)
    V.step(dst, 0.7, True, src)
    got = dst.download()
    assert rel(got, D0 + 0.7 * Y) < tol
    dst.upload(D0)
    V.step(dst, 0.7, True, src)
    assert np.array_equal(dst.download(), got)
