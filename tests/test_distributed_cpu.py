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
