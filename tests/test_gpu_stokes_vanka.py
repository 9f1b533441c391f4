"""Cell-patch Vanka smoother of the two-variable Stokes system (SURVEY 8 f-1 for BASELINE configs[4]; reference include/stmg.h:626-738,
832-872 as tests/tp_03stokes.cc:714-726 creates it): the HIP apply (stfem_stokes_vanka_*: class blocks read off the device operator on
1 - 3 cells per direction, MFMA apply + collecting launch) against the dense numpy restatement that restricts the assembled matrices
of the WHOLE mesh cell by cell (oracle/vanka_oracle.py::StokesVankaOracle).  fp64, rel-L2 <= 1e-10 (the blocks are inverted)."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(np.ravel(a) - np.ravel(b)) / max(np.linalg.norm(np.ravel(b)), 1e-300)


@pytest.fixture(scope="module")
def stfem():
    mod = importlib.import_module("dealii-stfem_amd")
    mod.lib()
    return mod


def _blocks(stfem, nt, ns, variable_major):
    nb = 2 * nt * ns
    var = [0] * nb
    for it in range(ns):
        for d in range(nt):
            var[stfem.stokes_block_index(nt, it, 1, d, variable_major)] = 1
    return var


@pytest.mark.parametrize("nc,ttype,r,ns,mask,weak,dg,upper,variable_major", [
    ((3, 3, 3), 0, 1, 1, 63, 0, False, (1.0, 1.0, 1.0), True),        # cG(1): one time dof, 89 rows; all 27 classes
    ((2, 3, 2), 1, 1, 1, 63, 0, False, (1.0, 1.5, 0.5), True),        # dG(1): two time dofs, 178 rows, anisotropic cells
    ((3, 2, 2), 0, 2, 1, 63 & ~3, 3, False, (1.0, 1.0, 1.0), False),  # cG(2), weak (Nitsche) x faces, time-major blocks
    ((2, 2, 3), 0, 1, 1, 63, 0, True, (1.0, 1.0, 1.0), True),         # FE_DGP(1) pressure: 85 rows
    ((2, 2, 2), 1, 1, 2, 63, 0, True, (1.0, 1.0, 1.0), True),         # dG(1), two time steps at once, DGP: 8 blocks, 340 rows
    ((1, 1, 1), 0, 1, 1, 0, 0, True, (1.0, 1.0, 1.0), True),          # a single unconstrained cell: the exact inverse of the system
])
def test_stokes_vanka_vs_oracle(nc, ttype, r, ns, mask, weak, dg, upper, variable_major, stfem):
    from oracle import vanka_oracle
    Alpha, Beta, _, _ = stfem.get_fe_time_weights_stokes(ttype, r, 0.05, ns)
    nt = r if ttype == 0 else r + 1
    nb = 2 * nt * ns
    var = _blocks(stfem, nt, ns, variable_major)
    if not variable_major:  # the matrices of get_fe_time_weights_stokes are in variable-major block order: permute
        perm = [0] * nb
        for it in range(ns):
            for v in range(2):
                for d in range(nt):
                    perm[stfem.stokes_block_index(nt, it, v, d, False)] = stfem.stokes_block_index(nt, it, v, d, True)
        Alpha, Beta = Alpha[np.ix_(perm, perm)], Beta[np.ix_(perm, perm)]
    verts = stfem.mesh_vertices(nc, (0, 0, 0), upper)
    op = stfem.StokesMatrixFreeOperator(nc, upper=upper, dirichlet_mask=mask, viscosity=0.7,
                                        weak_boundary_ids=[f for f in range(6) if weak >> f & 1], dg_pressure=dg)
    V = stfem.StokesPreconditionVanka(op, var, Alpha, Beta)
    ncls = 1
    for d in range(3):
        ncls *= min(nc[d], 3)
    assert V.n_classes == ncls
    ref = vanka_oracle.StokesVankaOracle(nc, verts, mask, 0.7, var, Alpha, Beta, weak_mask=weak, dg_pressure=dg)
    rng = np.random.default_rng(3)
    X = [rng.uniform(-1, 1, 3 * op.n_velocity if v == 0 else op.n_pressure) for v in var]
    src = [op.initialize_dof_vector(v, x) for v, x in zip(var, X)]
    dst = [op.initialize_dof_vector(v, np.full(x.size, 9.0)) for v, x in zip(var, X)]  # overwritten
    V.vmult(dst, src)
    want = ref.vmult(X)
    Y = [d.download() for d in dst]
    assert rel(np.concatenate(Y), np.concatenate(want)) < 1e-10
    for b in range(nb):
        assert rel(Y[b], want[b]) < 1e-9, b
    V.vmult(dst, src)  # reproducible
    assert all(np.array_equal(d.download(), y) for d, y in zip(dst, Y))
    # the relaxation step: dst += omega V src
    V.step(dst, 0.6, True, src)
    assert rel(np.concatenate([d.download() for d in dst]), 1.6 * np.concatenate(Y)) < 1e-12
    with pytest.raises(stfem.StfemError):
        V.vmult(src, src)


def test_stokes_vanka_relaxation_reduces_the_residual(stfem):
    """A few sweeps of x <- x + omega P^-1 (b - A x) with the smoother contract on the space-time Stokes system (what the multigrid
    levels of tests/tp_03stokes.cc use it for)."""
    nc, nt = (4, 4, 4), 1
    Alpha, Beta, _, _ = stfem.get_fe_time_weights_stokes(stfem.CGP, 1, 0.1, 1)
    op = stfem.StokesMatrixFreeOperator(nc, dirichlet_mask=63, viscosity=1.0, dg_pressure=True)
    var = [0, 1]
    V = stfem.StokesPreconditionVanka(op, var, Alpha, Beta)
    rng = np.random.default_rng(5)
    sizes = [3 * op.n_velocity, op.n_pressure]
    b = [op.initialize_dof_vector(v, rng.uniform(-1, 1, n)) for v, n in zip(var, sizes)]
    # zero the constrained velocity rows of b (the operator returns exact zeros there)
    t = [op.initialize_dof_vector(v) for v in var]
    one = [op.initialize_dof_vector(v, np.ones(n)) for v, n in zip(var, sizes)]
    op.st_vmult(Alpha, Beta, 1, nt, t, one)
    bu = b[0].download()
    probe = [op.initialize_dof_vector(0, np.ones(sizes[0])), op.initialize_dof_vector(1)]
    mu = op.initialize_dof_vector(0)
    op.mass_vmult(mu, probe[0])
    bu[mu.download() == 0.0] = 0.0
    b[0].upload(bu)
    x = [op.initialize_dof_vector(v) for v in var]
    r = [op.initialize_dof_vector(v) for v in var]
    bh = [v.download() for v in b]
    norms = []
    for it in range(8):
        op.st_vmult(Alpha, Beta, 1, nt, r, x)
        res = [bh[i] - r[i].download() for i in range(2)]
        norms.append(np.sqrt(sum(np.sum(q * q) for q in res)))
        for i in range(2):
            r[i].upload(res[i])
        V.step(x, 0.5, True, r)
    assert norms[-1] < 0.5 * norms[0], norms
