"""The N > 1 data path on ONE GPU: a mesh cut into z-slabs, every slab with its own context on
cuda:0 (open interface faces: Slab.dirichlet_mask), the HIP sweep on every slab, the interface
planes packed with stfem_plane_pack, copied device-to-device (what RCCL send/recv does between
GPUs) and added with stfem_plane_unpack(add = 1).  The result must equal the single-domain HIP
result and the CPU oracle: compress(add) + update_ghost_values around the reference's cell loop
(include/operators.h:1016-1017) in one packed exchange per space-time vmult.
Cartesian and perturbed meshes, fp64 and fp32, 2 and 3 slabs."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope="module")
def stfem():
    mod = importlib.import_module("dealii-stfem_amd")
    mod.lib()
    return mod


@pytest.mark.parametrize("p,gnc,world,distort,number", [
    (4, (7, 5, 6), 2, 0.0, "double"),
    (4, (7, 5, 6), 3, 0.0, "double"),
    (4, (4, 5, 6), 2, 0.15, "double"),
    (2, (6, 4, 7), 3, 0.15, "double"),
    (3, (5, 5, 4), 2, 0.0, "float"),
    (4, (4, 3, 4), 2, 0.15, "float"),
])
def test_slab_exchange_on_one_gpu(stfem, oracle_mod, p, gnc, world, distort, number):
    import ctypes
    dmod = importlib.import_module("dealii-stfem_amd.distributed")
    L = stfem.lib()
    hip = ctypes.CDLL("libamdhip64.so")  # the runtime the library itself uses (already loaded)
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipDeviceSynchronize.argtypes = []
    esz = 8 if number == "double" else 4
    tol = 1e-12 if number == "double" else 2e-5
    Alpha, Beta, _, _ = stfem.get_fe_time_weights(stfem.CGP, 2, 0.01, 1)
    nb = Alpha.shape[0]
    nx, ny = p * gnc[0] + 1, p * gnc[1] + 1
    plane = nx * ny
    upper = (1.0, 1.0, 1.5)
    gv = stfem.mesh_vertices(gnc, (0, 0, 0), upper, distort, 5489)
    rng = np.random.default_rng(11)
    ndofs = plane * (p * gnc[2] + 1)
    X = rng.uniform(-1, 1, (nb, ndofs))

    # single domain: HIP and oracle
    gctx = stfem.MatrixFreeOperator(p, gnc, vertices=gv, number=number)
    gA = stfem.SystemMatrix(gctx, Alpha, Beta)
    gdst = gA.initialize_dof_vector()
    gA.vmult(gdst, stfem.BlockVector(gctx, nb).upload(X))
    Ygpu = gdst.download()
    Yref = oracle_mod.Oracle(p, gnc, gv, 63).st_vmult(Alpha, Beta, X)
    assert rel(Ygpu, Yref) < tol

    # the slabs: local sweep, pack
    ranks = []
    for r in range(world):
        slab = dmod.make_slab(gnc, r, world)
        lv = stfem.mesh_vertices(gnc, (0, 0, 0), upper, distort, 5489, z_range=(slab.z0, slab.z1))
        ctx = stfem.MatrixFreeOperator(p, slab.ncell, vertices=lv, number=number,
                                       dirichlet_mask=slab.dirichlet_mask(63))
        A = stfem.SystemMatrix(ctx, Alpha, Beta)
        lo, hi = p * slab.z0 * plane, (p * slab.z1 + 1) * plane
        src = stfem.BlockVector(ctx, nb).upload(X[:, lo:hi])  # owned + ghost plane, consistent
        dst = A.initialize_dof_vector()
        A.vmult(dst, src)
        # packed-plane buffers (n_blocks planes each): device memory from the library itself
        nzl = p * (slab.z1 - slab.z0) + 1
        assert nzl >= nb
        hold = stfem.BlockVector(ctx, 4)
        bufs = {k: hold.block_ptr(q) for q, k in enumerate(("ts", "bs", "tr", "br"))}
        ranks.append(dict(slab=slab, ctx=ctx, A=A, src=src, dst=dst, bufs=bufs, hold=hold, nzl=nzl, lo=lo, hi=hi))
    for R in ranks:
        if R["slab"].has_upper:
            assert L.stfem_plane_pack(R["ctx"]._h, R["dst"]._h, R["nzl"] - 1, R["bufs"]["ts"], None) == 0
        if R["slab"].has_lower:
            assert L.stfem_plane_pack(R["ctx"]._h, R["dst"]._h, 0, R["bufs"]["bs"], None) == 0
    assert hip.hipDeviceSynchronize() == 0
    # the exchange (between GPUs: ncclSend / ncclRecv of exactly these buffers)
    D2D = 3  # hipMemcpyDeviceToDevice
    for r, R in enumerate(ranks):
        if R["slab"].has_upper:
            assert hip.hipMemcpy(ranks[r + 1]["bufs"]["br"], R["bufs"]["ts"], nb * plane * esz, D2D) == 0
        if R["slab"].has_lower:
            assert hip.hipMemcpy(ranks[r - 1]["bufs"]["tr"], R["bufs"]["bs"], nb * plane * esz, D2D) == 0
    assert hip.hipDeviceSynchronize() == 0
    for R in ranks:
        if R["slab"].has_upper:
            assert L.stfem_plane_unpack(R["ctx"]._h, R["dst"]._h, R["nzl"] - 1, R["bufs"]["tr"], 1, None) == 0
        if R["slab"].has_lower:
            assert L.stfem_plane_unpack(R["ctx"]._h, R["dst"]._h, 0, R["bufs"]["br"], 1, None) == 0
    assert hip.hipDeviceSynchronize() == 0
    for R in ranks:
        Y = R["dst"].download()
        assert rel(Y, Ygpu[:, R["lo"]:R["hi"]]) < tol  # owner and ghost copies of every interface plane agree
        assert rel(Y, Yref[:, R["lo"]:R["hi"]]) < tol

    # argument checks of the pack / unpack entry points (include/stfem.h)
    R = ranks[0]
    assert L.stfem_plane_pack(R["ctx"]._h, R["dst"]._h, R["nzl"], R["bufs"]["ts"], None) != 0
    assert L.stfem_plane_unpack(R["ctx"]._h, R["dst"]._h, -1, R["bufs"]["ts"], 1, None) != 0
    assert L.stfem_plane_pack(R["ctx"]._h, R["dst"]._h, 0, None, None) != 0


@pytest.mark.parametrize("p,gnc,world,number", [(2, (4, 3, 6), 2, "double"), (4, (3, 2, 6), 3, "double"), (3, (2, 3, 4), 2, "float")])
def test_partitioned_vanka_on_one_gpu(stfem, p, gnc, world, number):
    """The cell-patch Vanka smoother on z-slabs (stfem_vanka_create_partitioned: the cells behind an interface face count in blocks and
    valences, the apply leaves partial sums in the interface planes) + the add-exchange of the planes == the smoother of the whole mesh."""
    import ctypes
    dmod = importlib.import_module("dealii-stfem_amd.distributed")
    L = stfem.lib()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    esz = 8 if number == "double" else 4
    tol = 1e-12 if number == "double" else 2e-5
    Alpha, Beta, _, _ = stfem.get_fe_time_weights(stfem.CGP, 2, 0.05, 1)
    nb = Alpha.shape[0]
    plane = (p * gnc[0] + 1) * (p * gnc[1] + 1)
    upper = (1.0, 1.0, 1.5)
    ndofs = plane * (p * gnc[2] + 1)
    X = np.random.default_rng(3).uniform(-1, 1, (nb, ndofs))
    if number == "float":
        X = X.astype(np.float32).astype(float)
    gctx = stfem.MatrixFreeOperator(p, gnc, upper=upper, number=number)
    gV = stfem.PreconditionVanka(gctx, Alpha, Beta)
    gdst = stfem.BlockVector(gctx, nb)
    gV.vmult(gdst, stfem.BlockVector(gctx, nb).upload(X))
    Y = gdst.download()
    ranks = []
    for r in range(world):
        slab = dmod.make_slab(gnc, r, world)
        lo_z, hi_z = upper[2] * slab.z0 / gnc[2], upper[2] * slab.z1 / gnc[2]
        ctx = stfem.MatrixFreeOperator(p, slab.ncell, lower=(0, 0, lo_z), upper=(upper[0], upper[1], hi_z), number=number,
                                       dirichlet_mask=slab.dirichlet_mask(63))
        nmask = (16 if slab.has_lower else 0) | (32 if slab.has_upper else 0)
        V = stfem.PreconditionVanka(ctx, Alpha, Beta, neighbour_mask=nmask)
        lo, hi = p * slab.z0 * plane, (p * slab.z1 + 1) * plane
        dst = stfem.BlockVector(ctx, nb)
        V.vmult(dst, stfem.BlockVector(ctx, nb).upload(X[:, lo:hi]))
        hold = stfem.BlockVector(ctx, 4)
        nzl = p * (slab.z1 - slab.z0) + 1
        ranks.append(dict(slab=slab, ctx=ctx, V=V, dst=dst, hold=hold, nzl=nzl, lo=lo, hi=hi,
                          bufs={k: hold.block_ptr(q) for q, k in enumerate(("ts", "bs", "tr", "br"))}))
    for R in ranks:
        if R["slab"].has_upper:
            assert L.stfem_plane_pack(R["ctx"]._h, R["dst"]._h, R["nzl"] - 1, R["bufs"]["ts"], None) == 0
        if R["slab"].has_lower:
            assert L.stfem_plane_pack(R["ctx"]._h, R["dst"]._h, 0, R["bufs"]["bs"], None) == 0
    assert hip.hipDeviceSynchronize() == 0
    for r, R in enumerate(ranks):
        if R["slab"].has_upper:
            assert hip.hipMemcpy(ranks[r + 1]["bufs"]["br"], R["bufs"]["ts"], nb * plane * esz, 3) == 0
        if R["slab"].has_lower:
            assert hip.hipMemcpy(ranks[r - 1]["bufs"]["tr"], R["bufs"]["bs"], nb * plane * esz, 3) == 0
    for R in ranks:
        if R["slab"].has_upper:
            assert L.stfem_plane_unpack(R["ctx"]._h, R["dst"]._h, R["nzl"] - 1, R["bufs"]["tr"], 1, None) == 0
        if R["slab"].has_lower:
            assert L.stfem_plane_unpack(R["ctx"]._h, R["dst"]._h, 0, R["bufs"]["br"], 1, None) == 0
    assert hip.hipDeviceSynchronize() == 0
    for R in ranks:
        assert rel(R["dst"].download(), Y[:, R["lo"]:R["hi"]]) < tol
    # a mask naming a Dirichlet face, and per-cell contexts, are refused
    with pytest.raises(stfem.StfemError):
        stfem.PreconditionVanka(gctx, Alpha, Beta, neighbour_mask=16)
    pert = stfem.MatrixFreeOperator(p, (2, 2, 2), vertices=stfem.mesh_vertices((2, 2, 2), distort=0.1), dirichlet_mask=63 & ~32, number=number)
    with pytest.raises(stfem.StfemError):
        stfem.PreconditionVanka(pert, Alpha, Beta, neighbour_mask=32)


def _exchange_add(stfem, ranks, nb, plane, esz):
    """the add-exchange of the interface planes between slab contexts on one device: pack -> device copy -> unpack-add"""
    import ctypes
    L = stfem.lib()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    for R in ranks:
        if R["slab"].has_upper:
            assert L.stfem_plane_pack(R["ctx"]._h, R["dst"]._h, R["nzl"] - 1, R["bufs"]["ts"], None) == 0
        if R["slab"].has_lower:
            assert L.stfem_plane_pack(R["ctx"]._h, R["dst"]._h, 0, R["bufs"]["bs"], None) == 0
    assert hip.hipDeviceSynchronize() == 0
    for r, R in enumerate(ranks):
        if R["slab"].has_upper:
            assert hip.hipMemcpy(ranks[r + 1]["bufs"]["br"], R["bufs"]["ts"], nb * plane * esz, 3) == 0
        if R["slab"].has_lower:
            assert hip.hipMemcpy(ranks[r - 1]["bufs"]["tr"], R["bufs"]["bs"], nb * plane * esz, 3) == 0
    for R in ranks:
        if R["slab"].has_upper:
            assert L.stfem_plane_unpack(R["ctx"]._h, R["dst"]._h, R["nzl"] - 1, R["bufs"]["tr"], 1, None) == 0
        if R["slab"].has_lower:
            assert L.stfem_plane_unpack(R["ctx"]._h, R["dst"]._h, 0, R["bufs"]["br"], 1, None) == 0
    assert hip.hipDeviceSynchronize() == 0


@pytest.mark.parametrize("p,gnc,world,number,coef", [(2, (4, 3, 6), 2, "double", False), (4, (3, 2, 6), 3, "double", False), (3, (3, 2, 5), 2, "float", False),
                                                      (2, (3, 3, 7), 3, "double", True)])
def test_partitioned_per_cell_vanka_on_one_gpu(stfem, p, gnc, world, number, coef):
    """One Vanka block per cell on z-slabs of a PERTURBED mesh (BASELINE configs[2] on several ranks;
    stfem_vanka_create_partitioned_general): the blocks of the cells next to an interface are built with the neighbour rank's cells
    from a context that carries one ghost cell layer (the reference builds them on owned and ghost cells, stmg.h:688-689, 795-796);
    slab smoother + add-exchange == the smoother of the whole mesh."""
    dmod = importlib.import_module("dealii-stfem_amd.distributed")
    esz = 8 if number == "double" else 4
    tol = 1e-11 if number == "double" else 5e-5
    Alpha, Beta, _, _ = stfem.get_fe_time_weights(stfem.CGP, 2, 0.05, 1)
    nb = Alpha.shape[0]
    plane = (p * gnc[0] + 1) * (p * gnc[1] + 1)
    ndofs = plane * (p * gnc[2] + 1)
    cpl = gnc[0] * gnc[1]
    X = np.random.default_rng(3).uniform(-1, 1, (nb, ndofs))
    if number == "float":
        X = X.astype(np.float32).astype(float)
    verts = lambda z0, z1: stfem.mesh_vertices(gnc, (0, 0, 0), (1, 1, 1.5), 0.15, 5489, z_range=(z0, z1))  # noqa: E731
    cvals = np.random.default_rng(8).uniform(0.5, 3.0, cpl * gnc[2])  # a cell-wise coefficient on K (then the extended context needs it too)
    gctx = stfem.MatrixFreeOperator(p, gnc, vertices=verts(0, gnc[2]), number=number)
    if coef:
        gctx.evaluate_coefficient(cvals, which=1)
    gV = stfem.PreconditionVanka(gctx, Alpha, Beta)
    assert gV.n_classes == gctx.n_cells  # one block per cell
    gdst = stfem.BlockVector(gctx, nb)
    gV.vmult(gdst, stfem.BlockVector(gctx, nb).upload(X))
    Y = gdst.download()
    ranks = []
    for r in range(world):
        slab = dmod.make_slab(gnc, r, world)
        ctx = stfem.MatrixFreeOperator(p, slab.ncell, vertices=verts(slab.z0, slab.z1), number=number, dirichlet_mask=slab.dirichlet_mask(63))
        e0, e1 = slab.z0 - (1 if slab.has_lower else 0), slab.z1 + (1 if slab.has_upper else 0)
        ext = stfem.MatrixFreeOperator(p, (gnc[0], gnc[1], e1 - e0), vertices=verts(e0, e1), number=number, dirichlet_mask=slab.dirichlet_mask(63))
        if coef:
            ctx.evaluate_coefficient(cvals[cpl * slab.z0:cpl * slab.z1], which=1)
            ext.evaluate_coefficient(cvals[cpl * e0:cpl * e1], which=1)
        nmask = (16 if slab.has_lower else 0) | (32 if slab.has_upper else 0)
        V = stfem.PreconditionVanka(ctx, Alpha, Beta, neighbour_mask=nmask, extended=ext)
        del ext  # (may be destroyed after the call)
        lo, hi = p * slab.z0 * plane, (p * slab.z1 + 1) * plane
        dst = stfem.BlockVector(ctx, nb)
        V.vmult(dst, stfem.BlockVector(ctx, nb).upload(X[:, lo:hi]))
        hold = stfem.BlockVector(ctx, 4)
        ranks.append(dict(slab=slab, ctx=ctx, V=V, dst=dst, hold=hold, nzl=p * (slab.z1 - slab.z0) + 1, lo=lo, hi=hi,
                          bufs={k: hold.block_ptr(q) for q, k in enumerate(("ts", "bs", "tr", "br"))}))
    _exchange_add(stfem, ranks, nb, plane, esz)
    for R in ranks:
        assert rel(R["dst"].download(), Y[:, R["lo"]:R["hi"]]) < tol
    # an extended context of the wrong shape is refused
    with pytest.raises(stfem.StfemError):
        stfem.PreconditionVanka(ranks[0]["ctx"], Alpha, Beta, neighbour_mask=32, extended=ranks[0]["ctx"])


@pytest.mark.parametrize("pf,pc,gf,gc,world,number", [(2, 2, (4, 4, 8), (2, 2, 4), 2, "double"), (4, 2, (2, 3, 6), (2, 3, 6), 3, "double"),
                                                       (3, 3, (2, 2, 12), (1, 1, 6), 3, "float")])
def test_partitioned_space_transfer_on_one_gpu(stfem, pf, pc, gf, gc, world, number):
    """Multigrid space transfer between the two levels of a z-slab (stfem_transfer_create_partitioned): the prolongation is local, the
    restriction + add-exchange of the coarse interface planes == the restriction of the whole mesh (the fine ghost plane counts once)."""
    import ctypes
    dmod = importlib.import_module("dealii-stfem_amd.distributed")
    L = stfem.lib()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    esz = 8 if number == "double" else 4
    tol = 1e-13 if number == "double" else 2e-6
    nb = 2
    gfine, gcoarse = stfem.MatrixFreeOperator(pf, gf, number=number), stfem.MatrixFreeOperator(pc, gc, number=number)
    T = stfem.MGTwoLevelTransfer(gfine, gcoarse)
    rng = np.random.default_rng(9)
    Xf, Xc = rng.uniform(-1, 1, (nb, gfine.n_dofs)), rng.uniform(-1, 1, (nb, gcoarse.n_dofs))
    if number == "float":
        Xf, Xc = Xf.astype(np.float32).astype(float), Xc.astype(np.float32).astype(float)
    rf, pc_ = stfem.BlockVector(gcoarse, nb), stfem.BlockVector(gfine, nb)
    T.restrict_and_add(rf, stfem.BlockVector(gfine, nb).upload(Xf))
    T.prolongate(pc_, stfem.BlockVector(gcoarse, nb).upload(Xc))
    RX, PX = rf.download(), pc_.download()
    plane_f, plane_c = (pf * gf[0] + 1) * (pf * gf[1] + 1), (pc * gc[0] + 1) * (pc * gc[1] + 1)
    ranks = []
    for r in range(world):
        sf, sc = dmod.make_slab(gf, r, world), dmod.make_slab(gc, r, world)
        assert sf.z0 * gc[2] == sc.z0 * gf[2] and sf.z1 * gc[2] == sc.z1 * gf[2]
        zf = lambda z: float(z) / gf[2]  # noqa: E731
        fine = stfem.MatrixFreeOperator(pf, sf.ncell, lower=(0, 0, zf(sf.z0)), upper=(1, 1, zf(sf.z1)), number=number, dirichlet_mask=sf.dirichlet_mask(63))
        coarse = stfem.MatrixFreeOperator(pc, sc.ncell, lower=(0, 0, zf(sf.z0)), upper=(1, 1, zf(sf.z1)), number=number, dirichlet_mask=sc.dirichlet_mask(63))
        Tr = stfem.MGTwoLevelTransfer(fine, coarse, neighbour_mask=(16 if sf.has_lower else 0) | (32 if sf.has_upper else 0))
        lof, hif = pf * sf.z0 * plane_f, (pf * sf.z1 + 1) * plane_f
        loc, hic = pc * sc.z0 * plane_c, (pc * sc.z1 + 1) * plane_c
        # prolongation: local
        out_f = stfem.BlockVector(fine, nb)
        Tr.prolongate(out_f, stfem.BlockVector(coarse, nb).upload(Xc[:, loc:hic]))
        assert rel(out_f.download(), PX[:, lof:hif]) < tol
        dst = stfem.BlockVector(coarse, nb)
        Tr.restrict_and_add(dst, stfem.BlockVector(fine, nb).upload(Xf[:, lof:hif]))
        hold = stfem.BlockVector(coarse, 4)
        ranks.append(dict(slab=sc, ctx=coarse, fine=fine, T=Tr, dst=dst, hold=hold, nzl=pc * (sc.z1 - sc.z0) + 1, lo=loc, hi=hic,
                          bufs={k: hold.block_ptr(q) for q, k in enumerate(("ts", "bs", "tr", "br"))}))
    for R in ranks:
        if R["slab"].has_upper:
            assert L.stfem_plane_pack(R["ctx"]._h, R["dst"]._h, R["nzl"] - 1, R["bufs"]["ts"], None) == 0
        if R["slab"].has_lower:
            assert L.stfem_plane_pack(R["ctx"]._h, R["dst"]._h, 0, R["bufs"]["bs"], None) == 0
    assert hip.hipDeviceSynchronize() == 0
    for r, R in enumerate(ranks):
        if R["slab"].has_upper:
            assert hip.hipMemcpy(ranks[r + 1]["bufs"]["br"], R["bufs"]["ts"], nb * plane_c * esz, 3) == 0
        if R["slab"].has_lower:
            assert hip.hipMemcpy(ranks[r - 1]["bufs"]["tr"], R["bufs"]["bs"], nb * plane_c * esz, 3) == 0
    for R in ranks:
        if R["slab"].has_upper:
            assert L.stfem_plane_unpack(R["ctx"]._h, R["dst"]._h, R["nzl"] - 1, R["bufs"]["tr"], 1, None) == 0
        if R["slab"].has_lower:
            assert L.stfem_plane_unpack(R["ctx"]._h, R["dst"]._h, 0, R["bufs"]["br"], 1, None) == 0
    assert hip.hipDeviceSynchronize() == 0
    for R in ranks:
        assert rel(R["dst"].download(), RX[:, R["lo"]:R["hi"]]) < tol


class _SlabLevels:
    """A two- or three-level space multigrid on `world` z-slabs of one mesh, every slab with its own contexts on the same device and the
    interface planes exchanged by pack -> copy -> unpack-add: the operations a rank of the partitioned C++ mirror performs
    (host/stfem/stmg.h: level operator with its add-exchange, partitioned Vanka relaxation, local prolongation, restriction with partial
    sums in the coarse interface planes).  world = 1 is the whole mesh through the same code."""

    def __init__(self, stfem, gnc_fine, degrees, world, distort, Alpha, Beta, number="double"):
        self.stfem, self.dmod = stfem, importlib.import_module("dealii-stfem_amd.distributed")
        self.nb, self.world, self.esz = Alpha.shape[0], world, 8 if number == "double" else 4
        self.levels = []  # coarsest first: list over levels of list over ranks
        n_levels = len(degrees)
        for l in range(n_levels):
            f = 2 ** (n_levels - 1 - l)  # cells of this level are f fine cells wide
            gnc = tuple(c // f for c in gnc_fine)
            p = degrees[l]
            plane = (p * gnc[0] + 1) * (p * gnc[1] + 1)

            def verts(z0, z1, f=f, gnc=gnc):  # every f-th vertex plane / row / column of the fine mesh
                v = stfem.mesh_vertices(gnc_fine, (0, 0, 0), (1, 1, 1.5), distort, 5489, z_range=(f * z0, f * z1))
                v = v.reshape(f * (z1 - z0) + 1, gnc_fine[1] + 1, gnc_fine[0] + 1, 3)[::f, ::f, ::f]
                return np.ascontiguousarray(v).reshape(-1, 3)
            ranks = []
            for r in range(world):
                slab = self.dmod.make_slab(gnc, r, world)
                ctx = stfem.MatrixFreeOperator(p, slab.ncell, vertices=verts(slab.z0, slab.z1), number=number, dirichlet_mask=slab.dirichlet_mask(63))
                nmask = (16 if slab.has_lower else 0) | (32 if slab.has_upper else 0)
                ext = None
                if nmask:
                    e0, e1 = slab.z0 - (1 if slab.has_lower else 0), slab.z1 + (1 if slab.has_upper else 0)
                    ext = stfem.MatrixFreeOperator(p, (gnc[0], gnc[1], e1 - e0), vertices=verts(e0, e1), number=number, dirichlet_mask=slab.dirichlet_mask(63))
                hold = stfem.BlockVector(ctx, 4 if self.nb <= 4 else self.nb * 4)
                ranks.append(dict(slab=slab, ctx=ctx, A=stfem.SystemMatrix(ctx, Alpha, Beta), V=stfem.PreconditionVanka(ctx, Alpha, Beta, neighbour_mask=nmask, extended=ext),
                                  nmask=nmask, plane=plane, nzl=p * (slab.z1 - slab.z0) + 1, lo=p * slab.z0 * plane, hi=(p * slab.z1 + 1) * plane, hold=hold,
                                  bufs={k: hold.block_ptr(q) for q, k in enumerate(("ts", "bs", "tr", "br"))}))
            self.levels.append(ranks)
        self.T = [None] + [[stfem.MGTwoLevelTransfer(fr["ctx"], cr["ctx"], neighbour_mask=fr["nmask"]) for fr, cr in zip(self.levels[l], self.levels[l - 1])]
                           for l in range(1, n_levels)]

    def vec(self, l):
        return [self.stfem.BlockVector(R["ctx"], self.nb) for R in self.levels[l]]

    def compress(self, l, v):
        if self.world > 1:
            ranks = [dict(R, dst=x) for R, x in zip(self.levels[l], v)]
            _exchange_add(self.stfem, ranks, self.nb, self.levels[l][0]["plane"], self.esz)

    def axpby(self, l, a, x, b, y):
        L = self.stfem.lib()
        for R, xi, yi in zip(self.levels[l], x, y):
            assert L.stfem_vector_axpby(R["ctx"]._h, a, xi._h, b, yi._h, None) == 0

    def vmult(self, l, dst, src):
        for R, d, s in zip(self.levels[l], dst, src):
            R["A"].vmult(d, s)
        self.compress(l, dst)

    def relax(self, l, dst, src, omega):  # dst = omega P^-1 src
        for R, d, s in zip(self.levels[l], dst, src):
            R["V"].vmult(d, s)
        self.compress(l, dst)
        self.axpby(l, 0.0, dst, omega, dst)

    def v_cycle(self, l, u, d, omega):
        """Multigrid::level_v_step with one relaxation step per level (from zero before, one more after the coarse correction)"""
        self.relax(l, u, d, omega)
        if l == 0:
            return
        t, dc, uc = self.vec(l), self.vec(l - 1), self.vec(l - 1)
        self.vmult(l, t, u)
        self.axpby(l, 1.0, d, -1.0, t)                     # t = d - A u
        for T, c, f in zip(self.T[l], dc, t):               # restriction: partial sums in the coarse interface planes
            T.restrict_and_add(c, f)
        self.compress(l - 1, dc)
        self.v_cycle(l - 1, uc, dc, omega)
        for T, f, c in zip(self.T[l], t, uc):               # prolongation: local
            T.prolongate(f, c)
        self.axpby(l, 1.0, t, 1.0, u)
        self.vmult(l, t, u)
        self.axpby(l, 1.0, d, -1.0, t)
        r = self.vec(l)
        self.relax(l, r, t, omega)
        self.axpby(l, 1.0, r, 1.0, u)


@pytest.mark.parametrize("gnc,degrees,world,distort", [((4, 4, 8), (2, 2), 2, 0.0), ((4, 4, 12), (1, 2, 2), 3, 0.0), ((4, 4, 8), (2, 2), 2, 0.12),
                                                       ((4, 4, 12), (2, 2, 2), 3, 0.1)])
def test_v_cycle_on_slabs_equals_whole_mesh(stfem, gnc, degrees, world, distort):
    """One full V-cycle (level operators, Vanka relaxation, restriction, prolongation) on 2 and 3 z-slabs with the interface planes
    exchanged == the same cycle on the whole mesh: uniform meshes (block classes) and perturbed ones (one block per cell, ghost cell
    layers: the configs[2] situation on several ranks).  Every level halves the cells; degrees (1, 2, 2): the coarsest transfer
    changes mesh and degree at once."""
    Alpha, Beta, _, _ = stfem.get_fe_time_weights(stfem.CGP, 2, 0.05, 1)
    nb = Alpha.shape[0]
    whole = _SlabLevels(stfem, gnc, degrees, 1, distort, Alpha, Beta)
    parts = _SlabLevels(stfem, gnc, degrees, world, distort, Alpha, Beta)
    top = len(degrees) - 1
    n = whole.levels[top][0]["ctx"].n_dofs
    X = np.random.default_rng(12).uniform(-1, 1, (nb, n))
    # a right-hand side of the constrained space, as FGMRES hands one over: A x
    x, d, u = whole.vec(top), whole.vec(top), whole.vec(top)
    x[0].upload(X)
    whole.vmult(top, d, x)
    D = d[0].download()
    whole.v_cycle(top, u, d, 0.7)
    U = u[0].download()
    assert np.linalg.norm(U) > 0
    dp, up = parts.vec(top), parts.vec(top)
    for R, v in zip(parts.levels[top], dp):
        v.upload(D[:, R["lo"]:R["hi"]])
    parts.v_cycle(top, up, dp, 0.7)
    for R, v in zip(parts.levels[top], up):
        assert rel(v.download(), U[:, R["lo"]:R["hi"]]) < 1e-10


@pytest.mark.parametrize("p,nc,number,distort", [(2, (4, 3, 5), "double", 0.0), (4, (6, 2, 4), "double", 0.0), (3, (3, 3, 6), "double", 0.12),
                                                 (2, (5, 4, 3), "float", 0.1)])
def test_overlapped_slab_operator_equals_one_sweep(stfem, p, nc, number, distort):
    """dealii-stfem_amd/distributed.py::OverlappedSlabOperator (interface cell layers first, into vectors of their own; interior layers
    straight into the destination; stfem_planes_move): the same result as one sweep over the slab - what the exchange then completes."""
    dmod = importlib.import_module("dealii-stfem_amd.distributed")
    Alpha, Beta, _, _ = stfem.get_fe_time_weights(stfem.CGP, 2, 0.05, 1)
    nb = Alpha.shape[0]
    mask = 63 & ~48  # a slab in the middle of a partition: both z faces are interfaces
    verts = stfem.mesh_vertices(nc, distort=distort, seed=3) if distort else None
    nvx, nvy = nc[0] + 1, nc[1] + 1

    def make_ctx(z0, z1, m):
        snc = (nc[0], nc[1], z1 - z0)
        if verts is not None:
            v = np.asarray(verts).reshape(nc[2] + 1, nvy * nvx * 3)[z0:z1 + 1].reshape(-1)
            return stfem.MatrixFreeOperator(p, snc, vertices=v, number=number, dirichlet_mask=m)
        return stfem.MatrixFreeOperator(p, snc, lower=(0, 0, z0 / nc[2]), upper=(1, 1, z1 / nc[2]), number=number, dirichlet_mask=m)

    ctx = make_ctx(0, nc[2], mask)
    A = stfem.SystemMatrix(ctx, Alpha, Beta)
    rng = np.random.default_rng(6)
    X = rng.uniform(-1, 1, (nb, ctx.n_dofs))
    if number == "float":
        X = X.astype(np.float32).astype(np.float64)
    src, ref, dst = stfem.BlockVector(ctx, nb).upload(X), stfem.BlockVector(ctx, nb), stfem.BlockVector(ctx, nb)
    dst.upload(np.full((nb, ctx.n_dofs), 1e30))  # every entry must be written
    op = dmod.OverlappedSlabOperator(stfem, ctx, make_ctx, lambda c: stfem.SystemMatrix(c, Alpha, Beta), src, dst, mask)
    for transpose in (False, True):
        (A.Tvmult if transpose else A.vmult)(ref, src)
        op.vmult(transpose=transpose)
        want, got = ref.download(), dst.download()
        assert np.linalg.norm(got - want) <= (1e-14 if number == "double" else 1e-6) * np.linalg.norm(want)
