"""RCCL behind the C-ABI (include/stfem.h: stfem_comm_*, stfem_halo_begin/end, stfem_ghost_update,
stfem_dot_global) on ONE GPU: a one-rank communicator whose lower and upper neighbour are the rank
itself.  Every send is matched by the rank's own receive of the same position, so after the exchange
the top plane has received the top partial and the bottom plane the bottom partial: both interface
planes double, everything else is untouched - the data path pack -> ncclSend/ncclRecv on the
communicator's stream -> unpack-add, with the event hand-over between the two streams, is exercised
end to end.  (Two ranks on one GPU are refused by RCCL; the two-rank composition is covered with
device copies in tests/test_gpu_halo.py and with gloo in tests/test_distributed_cpu.py.)"""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("number", ["double", "float"])
def test_self_loop_exchange(number):
    stfem = importlib.import_module("dealii-stfem_amd")
    dmod = importlib.import_module("dealii-stfem_amd.distributed")
    p, nc, nb = 3, (4, 3, 5), 2
    ctx = stfem.MatrixFreeOperator(p, nc, number=number, dirichlet_mask=63 & ~48)
    comm = dmod.Communicator(0, 1, 0, lambda raw: raw)
    assert (comm.rank, comm.world) == (0, 1)
    nx, ny, nz = (p * c + 1 for c in nc)
    plane = nx * ny
    rng = np.random.default_rng(3)
    X = rng.uniform(-1, 1, (nb, ctx.n_dofs))
    if number == "float":
        X = X.astype(np.float32).astype(np.float64)
    v = stfem.BlockVector(ctx, nb).upload(X)
    comm.halo_begin(ctx, v, 0, 0)
    comm.halo_end(ctx, v)
    Y = v.download()
    ref = X.copy()
    ref[:, :plane] *= 2
    ref[:, -plane:] *= 2
    assert np.array_equal(Y, ref)

    # update_ghost_values: the top plane <- the (own) bottom plane; only the top plane changes
    comm.ghost_update(ctx, v, 0, 0)
    Z = v.download()
    ref2 = ref.copy()
    ref2[:, -plane:] = ref[:, :plane]
    assert np.array_equal(Z, ref2)

    # no neighbours: nothing moves; a second begin without end is refused
    comm.halo_begin(ctx, v, -1, -1)
    with pytest.raises(RuntimeError):
        comm.halo_begin(ctx, v, -1, -1)
    comm.halo_end(ctx, v)
    assert np.array_equal(v.download(), Z)
    with pytest.raises(RuntimeError):
        comm.halo_end(ctx, v)  # nothing in flight
    with pytest.raises(RuntimeError):
        comm.halo_begin(ctx, v, 0, 1)  # rank 1 does not exist

    # the reducing inner product (one rank: the local value)
    w = stfem.BlockVector(ctx, nb).upload(X)
    n_own = ctx.n_dofs - plane
    got = comm.dot(ctx, v, w, n_own)
    want = float(np.sum(Z[:, :n_own] * X[:, :n_own]))
    assert abs(got - want) <= (1e-12 if number == "double" else 1e-5) * abs(want)
    comm.close()


@pytest.mark.parametrize("number", ["double", "float"])
def test_cpp_caller_runs_partitioned(number, tmp_path, oracle_mod):
    """host/stfem/operators.h: set_partition + vmult + dot from C++ (no Python between operator and RCCL)."""
    import os
    import subprocess
    stfem = importlib.import_module("dealii-stfem_amd")
    host = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dealii-stfem_amd", "host")
    exe = os.path.join(host, "test_host_sharded")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", host], stdout=subprocess.DEVNULL)
    p, nc = 3, (3, 4, 3)
    out = tmp_path / "sh.bin"
    res = subprocess.run([exe, str(p), *map(str, nc), str(out), number], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    nb, n = (int(v) for v in np.fromfile(out, dtype=np.uint64, count=2))
    xy = float(np.fromfile(out, dtype=np.float64, count=1, offset=16)[0])
    X0, X1, Y, Z = np.fromfile(out, dtype=np.float64, offset=24).reshape(4, nb, n)
    plane = (p * nc[0] + 1) * (p * nc[1] + 1)
    tol = 1e-12 if number == "double" else 2e-5
    # update_ghost_values: the top plane of src became the bottom plane
    want_x = X0.copy()
    want_x[:, -plane:] = X0[:, :plane]
    if number == "float":
        want_x = want_x.astype(np.float32).astype(np.float64)
    assert np.array_equal(X1, want_x)
    Alpha, Beta, _, _ = stfem.get_fe_time_weights(stfem.CGP, 2, 1.0 / 32, 1)
    ref = oracle_mod.Oracle(p, nc, stfem.mesh_vertices(nc), 63 & ~48).st_vmult(Alpha, Beta, want_x)
    ref[:, :plane] *= 2
    ref[:, -plane:] *= 2
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)  # noqa: E731
    assert rel(Y, ref) < tol
    assert rel(Z, 2 * ref) < tol
    n_own = n - plane
    want = float(np.sum(X1[:, :n_own] * Y[:, :n_own]))
    assert abs(xy - want) <= 10 * tol * abs(want)


@pytest.mark.parametrize("distort", [0.0, 0.1])
def test_overlapped_exchange_on_the_self_loop(distort):
    """OverlappedSlabOperator with the RCCL exchange in the middle (stfem_halo_begin_split on the one-layer vectors before the interior
    sweep, stfem_halo_end on the assembled vector after it), on the self-loop communicator: the same vector as sweep, then
    stfem_halo_begin / end."""
    stfem = importlib.import_module("dealii-stfem_amd")
    dmod = importlib.import_module("dealii-stfem_amd.distributed")
    p, nc = 3, (4, 3, 5)
    mask = 63 & ~48
    Alpha, Beta, _, _ = stfem.get_fe_time_weights(stfem.CGP, 2, 0.05, 1)
    nb = Alpha.shape[0]
    verts = stfem.mesh_vertices(nc, distort=distort, seed=3) if distort else None

    def make_ctx(z0, z1, m):
        snc = (nc[0], nc[1], z1 - z0)
        if verts is not None:
            v = np.asarray(verts).reshape(nc[2] + 1, -1)[z0:z1 + 1].reshape(-1)
            return stfem.MatrixFreeOperator(p, snc, vertices=v, dirichlet_mask=m)
        return stfem.MatrixFreeOperator(p, snc, lower=(0, 0, z0 / nc[2]), upper=(1, 1, z1 / nc[2]), dirichlet_mask=m)

    ctx = make_ctx(0, nc[2], mask)
    comm = dmod.Communicator(0, 1, 0, lambda raw: raw)
    A = stfem.SystemMatrix(ctx, Alpha, Beta)
    rng = np.random.default_rng(12)
    src = stfem.BlockVector(ctx, nb).upload(rng.uniform(-1, 1, (nb, ctx.n_dofs)))
    ref, dst = stfem.BlockVector(ctx, nb), stfem.BlockVector(ctx, nb)
    A.vmult(ref, src)
    comm.halo_begin(ctx, ref, 0, 0)
    comm.halo_end(ctx, ref)
    op = dmod.OverlappedSlabOperator(stfem, ctx, make_ctx, lambda c: stfem.SystemMatrix(c, Alpha, Beta), src, dst, mask)
    for _ in range(2):  # (twice: the buffers and events of the communicator are reused)
        op.vmult(comm, 0, 0)
        want, got = ref.download(), dst.download()
        assert np.linalg.norm(got - want) <= 1e-14 * np.linalg.norm(want)
