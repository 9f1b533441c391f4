"""Cell-patch Vanka smoother: the HIP apply (stfem_vanka_vmult: MFMA GEMM per block class with
fused gather / scatter, blocks built from Kronecker products of restricted 1D matrices) against the dense
numpy restatement of the reference's PreconditionVanka (oracle/vanka_oracle.py, stmg.h:619-907)."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b)


@pytest.mark.parametrize("number", ["double", "float"])
@pytest.mark.parametrize("p,nc,ttype,r,mask,upper", [
    (2, (3, 3, 3), 0, 2, 63, (1.0, 1.0, 1.0)),
    (1, (5, 4, 3), 1, 0, 63, (1.0, 2.0, 0.5)),   # one temporal block, anisotropic cells
    (3, (4, 2, 3), 1, 1, 63 & ~48, (1.0, 1.0, 1.0)),  # dG(1), open z faces
    (4, (3, 2, 2), 0, 2, 63, (1.0, 1.0, 1.0)),   # cfg 1's element: 250 x 250 blocks
    (2, (1, 1, 1), 0, 2, 63, (1.0, 1.0, 1.0)),   # a single cell: the exact inverse
    (2, (2, 1, 4), 0, 3, 63 & ~3, (1.0, 1.0, 1.0)),  # three temporal blocks, 81 rows -> padded tiles
])
def test_vanka_vs_oracle(p, nc, ttype, r, mask, upper, number):
    from oracle import vanka_oracle
    stfem = importlib.import_module("dealii-stfem_amd")
    Alpha, Beta, _, _ = stfem.get_fe_time_weights(ttype, r, 0.05, 1)
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


def test_vanka_unsupported_contexts():
    stfem = importlib.import_module("dealii-stfem_amd")
    nc = (2, 2, 2)
    Alpha, Beta, _, _ = stfem.get_fe_time_weights(stfem.CGP, 2, 0.05, 1)
    ctx = stfem.MatrixFreeOperator(2, nc, vertices=stfem.mesh_vertices(nc, distort=0.1, seed=3))
    with pytest.raises(stfem.StfemError):
        stfem.PreconditionVanka(ctx, Alpha, Beta)  # per-cell blocks are not built yet
