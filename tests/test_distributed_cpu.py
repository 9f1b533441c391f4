"""N > 1 path on CPU: z-slab sharding + packed interface-plane exchange over torch.distributed
(gloo), with the oracle standing in for the HIP cell sweep.  Checks that the sharded result equals
the single-domain result, i.e. that one exchange per space-time vmult replaces the reference's
ghost update + compress(add) around every spatial cell loop (include/operators.h:1016-1017)."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, p, gnc, distort, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        stfem = importlib.import_module("dealii-stfem_amd")
        dmod = importlib.import_module("dealii-stfem_amd.distributed")
        from oracle import oracle
        oracle.lib().stfo_set_threads(2)
        Alpha, Beta, _, _ = stfem.get_fe_time_weights(stfem.CGP, 2, 0.01, 1)
        nb = Alpha.shape[0]
        nx, ny = p * gnc[0] + 1, p * gnc[1] + 1
        plane = nx * ny
        # global reference (every rank computes it: small)
        gv = stfem.mesh_vertices(gnc, (0, 0, 0), (1, 1, 2), distort, 5489)
        gorc = oracle.Oracle(p, gnc, gv, 63)
        X = np.stack([np.random.default_rng(7 + b).uniform(-1, 1, gorc.n_dofs) for b in range(nb)])
        Yref = gorc.st_vmult(Alpha, Beta, X)
        # this rank's slab
        slab = dmod.make_slab(gnc, rank, world)
        lv = stfem.mesh_vertices(gnc, (0, 0, 0), (1, 1, 2), distort, 5489, z_range=(slab.z0, slab.z1))
        lorc = oracle.Oracle(p, slab.ncell, lv, slab.dirichlet_mask(63))
        lo, hi = p * slab.z0 * plane, (p * slab.z1 + 1) * plane
        Xl = X[:, lo:hi].copy()  # owned + ghost plane, consistent by construction
        state = {}

        def local_vmult():
            state["dst"] = lorc.st_vmult(Alpha, Beta, Xl)

        def pack(iz, buf):
            iz = iz % (p * (slab.z1 - slab.z0) + 1)
            buf.copy_(torch.from_numpy(state["dst"][:, iz * plane:(iz + 1) * plane].reshape(-1).copy()))

        def unpack_add(iz, buf):
            iz = iz % (p * (slab.z1 - slab.z0) + 1)
            state["dst"][:, iz * plane:(iz + 1) * plane] += buf.numpy().reshape(nb, plane)

        bufs = {k: torch.zeros(nb * plane, dtype=torch.float64) for k in ("ts", "bs", "tr", "br")}
        dmod.sharded_vmult(slab, local_vmult, pack, unpack_add, bufs, dist)
        err = np.linalg.norm(state["dst"] - Yref[:, lo:hi]) / np.linalg.norm(Yref[:, lo:hi])
        # global squared norm over owned planes only == single-domain norm (the all-reduce the
        # Krylov solver needs)
        own = slab.n_owned_planes(p) * plane
        s = torch.tensor([float(np.sum(state["dst"][:, :own] ** 2))], dtype=torch.float64)
        dist.all_reduce(s)
        nerr = abs(s.item() - float(np.sum(Yref ** 2))) / float(np.sum(Yref ** 2))
        out[rank] = (err, nerr)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,p,gnc,distort", [(2, 2, (3, 2, 4), 0.0), (2, 4, (2, 2, 3), 0.15),
                                                 (3, 1, (3, 3, 7), 0.1)])
def test_sharded_vmult_gloo(world, p, gnc, distort):
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29500 + (os.getpid() * 7 + world * 13 + p) % 2000
    mp.spawn(_worker, args=(world, port, p, gnc, distort, out), nprocs=world, join=True)
    assert len(out) == world
    for rank in range(world):
        err, nerr = out[rank]
        assert err < 1e-12, (rank, err)
        assert nerr < 1e-12, (rank, nerr)


def test_slab_partition():
    dmod = importlib.import_module("dealii-stfem_amd.distributed")
    slabs = [dmod.make_slab((72, 72, 576), r, 8) for r in range(8)]
    assert [s.z1 - s.z0 for s in slabs] == [72] * 8
    assert slabs[0].z0 == 0 and slabs[-1].z1 == 576
    assert all(a.z1 == b.z0 for a, b in zip(slabs, slabs[1:]))
    assert slabs[0].dirichlet_mask(63) == 63 & ~32 and slabs[3].dirichlet_mask(63) == 15
    assert slabs[7].dirichlet_mask(63) == 63 & ~16
    ragged = [dmod.make_slab((4, 4, 10), r, 3) for r in range(3)]
    assert [s.z1 - s.z0 for s in ragged] == [4, 3, 3]
    assert sum(s.n_owned_planes(2) for s in ragged) == 2 * 10 + 1
    with pytest.raises(ValueError):
        dmod.make_slab((4, 4, 2), 0, 3)


# ---- a two-level V-cycle on z-slabs over gloo ----------------------------------------------------------------------------------------
# The sequence the partitioned multigrid runs (host/stfem/stmg.h with MatrixFreeOperator::set_partition; on one GPU:
# tests/test_gpu_halo.py::test_v_cycle_on_slabs_equals_whole_mesh): every level operator, smoother and restriction works on the
# slab's own cells and leaves partial sums in the interface planes, completed by one add-exchange each; the fine ghost plane is
# restricted by its owner only; the prolongation needs no exchange.  Here with the oracles standing in for the HIP kernels and
# torch.distributed (gloo) as the transport: the slabs' result equals the single-domain V-cycle of oracle/stmg_oracle.py.
def _vcycle_worker(rank, world, port, p, gnc_c, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        stfem = importlib.import_module("dealii-stfem_amd")
        dmod = importlib.import_module("dealii-stfem_amd.distributed")
        from oracle import oracle, stmg_oracle as mg, vanka_oracle
        oracle.lib().stfo_set_threads(2)
        Alpha, Beta, _, _ = stfem.get_fe_time_weights(stfem.CGP, 2, 0.05, 1)
        nb = Alpha.shape[0]
        omega, sweeps = 0.6, 2
        gnc = {0: tuple(gnc_c), 1: tuple(2 * c for c in gnc_c)}
        zext = 2.0
        # ---- single-domain reference (every rank computes it: small)
        G = {}
        for l in (0, 1):
            nc = gnc[l]
            verts = stfem.mesh_vertices(nc, (0, 0, 0), (1, 1, zext))
            orc = oracle.Oracle(p, nc, verts, 63)
            vk = vanka_oracle.VankaOracle(p, nc, verts, 63, Alpha, Beta)
            A = np.kron(Alpha, orc.dense(laplace=1.0)) + np.kron(Beta, orc.dense(mass=1.0))
            G[l] = dict(nc=nc, orc=orc, vk=vk, A=A, N=orc.n_dofs, plane=(p * nc[0] + 1) * (p * nc[1] + 1))
        P1 = mg.space_prolongation(p, gnc[1], 63, p, gnc[0], 63)
        import scipy.sparse as sp
        P = sp.block_diag([P1] * nb).tocsr()
        levels = [dict(A=G[l]["A"], smoother=(lambda r, l=l: G[l]["vk"].vmult(r.reshape(nb, -1)).ravel()), omega=omega, n_iterations=sweeps) for l in (0, 1)]
        X = np.random.default_rng(21).uniform(-1, 1, nb * G[1]["N"])
        want = mg.Multigrid(levels, [None, (P, P.T.tocsr())], variable=True, steps=1).vmult(X).reshape(nb, -1)
        # ---- this rank's slabs on both levels (the coarse slab boundaries are the fine ones halved)
        S = {0: dmod.make_slab(gnc[0], rank, world)}
        S[1] = dmod.Slab(rank, world, 2 * S[0].z0, 2 * S[0].z1, gnc[1])
        L = {}
        for l in (0, 1):
            s, g = S[l], G[l]
            verts = stfem.mesh_vertices(g["nc"], (0, 0, 0), (1, 1, zext), z_range=(s.z0, s.z1))
            lo, hi = p * s.z0 * g["plane"], (p * s.z1 + 1) * g["plane"]
            cells = [c for c in range(len(g["vk"].cells)) if s.z0 <= c // (g["nc"][0] * g["nc"][1]) < s.z1]
            L[l] = dict(orc=oracle.Oracle(p, s.ncell, verts, s.dirichlet_mask(63)), lo=lo, hi=hi, cells=cells, n=hi - lo)

        def exchange(l, v):
            """add-exchange of the two interface planes of the local block vector v [nb, n_local]"""
            s, pl = S[l], G[l]["plane"]
            ts, bs = torch.from_numpy(v[:, -pl:].copy().reshape(-1)), torch.from_numpy(v[:, :pl].copy().reshape(-1))
            tr, br = torch.zeros_like(ts), torch.zeros_like(bs)
            for w in dmod.exchange_add(s, ts, bs, tr, br, dist):
                w.wait()
            if s.has_upper:
                v[:, -pl:] += tr.numpy().reshape(nb, pl)
            if s.has_lower:
                v[:, :pl] += br.numpy().reshape(nb, pl)
            return v

        def apply_A(l, u):
            return exchange(l, L[l]["orc"].st_vmult(Alpha, Beta, u))

        def vanka(l, r):
            """the blocks of the slab's own cells (built with their neighbours on the other rank, as on owned + ghost cells)"""
            g, d = G[l], np.zeros_like(r)
            for c in L[l]["cells"]:
                idx = g["vk"].cells[c] - L[l]["lo"]
                d[:, idx] += (g["vk"].blocks[c] @ r[:, idx].ravel()).reshape(nb, -1)
            return exchange(l, d)

        def precondition(l, r):
            x = omega * vanka(l, r)
            for _ in range(1, sweeps):
                x = x + omega * vanka(l, r - apply_A(l, x))
            return x

        def smooth(l, u, rhs, from_zero, steps):
            i = 0
            if from_zero:
                u, i = precondition(l, rhs), 1
            for _ in range(i, steps):
                u = u + precondition(l, rhs - apply_A(l, u))
            return u

        Pl = P1[L[1]["lo"]:L[1]["hi"], L[0]["lo"]:L[0]["hi"]]  # the slab's rows and its coarse slab's columns

        def restrict(t):
            own = t.copy()
            if S[1].has_upper:
                own[:, -G[1]["plane"]:] = 0.0  # the fine ghost plane is restricted by its owner (the rank above)
            return exchange(0, (Pl.T @ own.T).T)

        defect = X.reshape(nb, -1)[:, L[1]["lo"]:L[1]["hi"]].copy()
        u = smooth(1, None, defect, True, 1)
        t = defect - apply_A(1, u)
        uc = smooth(0, None, restrict(t), True, 2)
        u = u + (Pl @ uc.T).T
        u = smooth(1, u, defect, False, 1)
        ref = want[:, L[1]["lo"]:L[1]["hi"]]
        out[rank] = float(np.linalg.norm(u - ref) / np.linalg.norm(ref))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,p,gnc_c", [(2, 2, (2, 2, 2)), (2, 1, (2, 3, 4))])
def test_v_cycle_on_slabs_gloo(world, p, gnc_c):
    mgr = mp.Manager()
    out = mgr.dict()
    port = 31500 + (os.getpid() * 11 + world * 17 + p) % 2000
    mp.spawn(_vcycle_worker, args=(world, port, p, gnc_c, out), nprocs=world, join=True)
    assert len(out) == world
    for rank in range(world):
        assert out[rank] < 1e-10, (rank, out[rank])
