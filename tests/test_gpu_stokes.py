"""GPU parity of the Stokes two-field operator through the C-ABI
(stfem_stokes_*): against the dense numpy fixtures and against the CPU oracle on a larger seeded
mesh.  Tolerance rel-L2 <= 1e-12 (fp64)."""
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-12
GOLD = os.path.join(os.path.dirname(__file__), "golden")
FIXTURES = ["stokes_cart_2x2x2", "stokes_pert_2x3x2", "stokes_free_3x2x2"]


def rel(a, b):
    return np.linalg.norm(np.ravel(a) - np.ravel(b)) / max(np.linalg.norm(np.ravel(b)), 1e-300)


@pytest.fixture(scope="module")
def stfem():
    mod = importlib.import_module("dealii-stfem_amd")
    mod.lib()
    return mod



@pytest.mark.parametrize("name", FIXTURES)
def test_stokes_golden_fixture(name, stfem):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    op = stfem.StokesMatrixFreeOperator(tuple(g["ncell"]), vertices=g["vertices"], dirichlet_mask=int(g["mask"]),
                                        viscosity=float(g["nu"]))
    nt = int(g["nt"])
    assert op.n_velocity == g["U"].shape[2] and op.n_pressure == g["P"].shape[1]
    for i in range(nt):
        u, p = op.initialize_dof_vector(0, g["U"][i]), op.initialize_dof_vector(1, g["P"][i])
        ou = op.initialize_dof_vector(0, np.full(3 * op.n_velocity, 7.0))   # vmult overwrites
        opr = op.initialize_dof_vector(1, np.full(op.n_pressure, -3.0))
        op.vmult(ou, opr, u, p)
        mu = op.initialize_dof_vector(0, np.full(3 * op.n_velocity, 5.0))
        op.mass_vmult(mu, u)
        assert rel(ou.download(), g["SU"][i]) < TOL
        assert rel(opr.download(), g["SP"][i]) < TOL
        assert rel(mu.download(), g["MU"][i]) < TOL
    # space-time: SystemMatrixStokes::vmult
    src = [None] * (2 * nt); dst = [None] * (2 * nt)
    for d in range(nt):
        src[stfem.stokes_block_index(nt, 0, 0, d)] = op.initialize_dof_vector(0, g["U"][d])
        src[stfem.stokes_block_index(nt, 0, 1, d)] = op.initialize_dof_vector(1, g["P"][d])
    for j in range(2 * nt):
        dst[j] = op.initialize_dof_vector(src[j].variable, np.full(src[j].size, 11.0))
    op.st_vmult(g["Alpha"], g["Beta"], 1, nt, dst, src)
    for d in range(nt):
        assert rel(dst[stfem.stokes_block_index(nt, 0, 0, d)].download(), g["DU"][d]) < TOL
        assert rel(dst[stfem.stokes_block_index(nt, 0, 1, d)].download(), g["DP"][d]) < TOL


@pytest.mark.parametrize("variable_major", [True, False])
def test_stokes_vs_oracle_two_steps(variable_major, stfem):
    """cG(2), two time steps at once (4 time dofs, 8 blocks), perturbed 5x4x6 mesh, against the
    reference-structured CPU oracle; both BlockSlice orderings."""
    from oracle import oracle
    nc = (5, 4, 6)
    verts = stfem.mesh_vertices(nc, distort=0.15, seed=77)
    nu, mask = 0.3, 0b111011
    op = stfem.StokesMatrixFreeOperator(nc, vertices=verts, dirichlet_mask=mask, viscosity=nu)
    orc = oracle.StokesOracle(nc, verts, mask, nu)
    assert (op.n_velocity, op.n_pressure) == (orc.n_u, orc.n_p)
    ns, r = 2, 2
    Alpha_vm, Beta_vm, _, _ = stfem.get_fe_time_weights_stokes(stfem.CGP, r, 1.0 / 16, ns)
    nt = r
    nb = 2 * nt * ns
    # permute to the requested block ordering
    perm = np.zeros(nb, dtype=int)
    for it in range(ns):
        for v in range(2):
            for d in range(nt):
                perm[stfem.stokes_block_index(nt, it, v, d, variable_major)] = stfem.stokes_block_index(nt, it, v, d, True)
    Alpha, Beta = Alpha_vm[np.ix_(perm, perm)], Beta_vm[np.ix_(perm, perm)]
    rng = np.random.default_rng(5)
    blocks = [None] * nb
    for it in range(ns):
        for d in range(nt):
            blocks[stfem.stokes_block_index(nt, it, 0, d, variable_major)] = rng.uniform(-1, 1, 3 * orc.n_u)
            blocks[stfem.stokes_block_index(nt, it, 1, d, variable_major)] = rng.uniform(-1, 1, orc.n_p)
    ref = orc.st_vmult(Alpha, Beta, ns, nt, blocks, variable_major)
    var = [0 if b.size == 3 * orc.n_u else 1 for b in blocks]
    src = [op.initialize_dof_vector(v, b) for v, b in zip(var, blocks)]
    dst = [op.initialize_dof_vector(v) for v in var]
    op.st_vmult(Alpha, Beta, ns, nt, dst, src, variable_major)
    for j in range(nb):
        assert rel(dst[j].download(), ref[j]) < TOL, j


def test_stokes_errors(stfem):
    op = stfem.StokesMatrixFreeOperator((2, 2, 2))
    u, p = op.initialize_dof_vector(0), op.initialize_dof_vector(1)
    with pytest.raises(stfem.StfemError) as e:
        op.vmult(u, p, u, p)
    assert e.value.status == -6  # STFEM_ERR_ALIAS, as deal.II forbids vmult(dst, src) with dst == src
    with pytest.raises(stfem.StfemError) as e:
        stfem.StokesMatrixFreeOperator((2, 2, 2), velocity_degree=3)
    assert e.value.status == -2  # STFEM_ERR_UNSUPPORTED


def test_stokes_vmult_slice_add(stfem):
    """n x 1 right-hand-side case (operators.h:748-781) against the oracle's spatial apply."""
    from oracle import oracle
    nc = (3, 4, 2)
    verts = stfem.mesh_vertices(nc, distort=0.1, seed=3)
    op = stfem.StokesMatrixFreeOperator(nc, vertices=verts, dirichlet_mask=63, viscosity=1.7)
    orc = oracle.StokesOracle(nc, verts, 63, 1.7)
    ns, nt = 2, 2
    nb = 2 * ns * nt
    rng = np.random.default_rng(11)
    Gamma, Zeta = rng.uniform(-1, 1, nb), rng.uniform(-1, 1, nb)
    Gamma[stfem.stokes_block_index(nt, 1, 1, 0)] = 0.0  # a skipped entry
    U, Pp = rng.uniform(-1, 1, 3 * orc.n_u), rng.uniform(-1, 1, orc.n_p)
    ku, kp = orc.apply(U, Pp)
    mu, _ = orc.apply(U, Pp, 0.0, 1.0)
    init = [rng.uniform(-1, 1, 3 * orc.n_u if (j // nt) % 2 == 0 else orc.n_p) for j in range(nb)]
    dst = [op.initialize_dof_vector((j // nt) % 2, init[j]) for j in range(nb)]
    op.st_vmult_slice_add(Gamma, Zeta, ns, nt, dst, op.initialize_dof_vector(0, U), op.initialize_dof_vector(1, Pp))
    for it in range(ns):
        for d in range(nt):
            ju, jp = stfem.stokes_block_index(nt, it, 0, d), stfem.stokes_block_index(nt, it, 1, d)
            assert rel(dst[ju].download(), init[ju] + Gamma[ju] * ku.reshape(-1) + Zeta[ju] * mu.reshape(-1)) < TOL
            assert rel(dst[jp].download(), init[jp] + Gamma[jp] * kp) < TOL


@pytest.mark.parametrize("nc,upper,mask", [((5, 4, 6), (1.0, 1.0, 1.0), 0b111011), ((3, 7, 2), (2.0, 0.5, 1.5), 63), ((21, 3, 3), (1.0, 1.0, 1.0), 0),
                                           ((1, 1, 1), (1.0, 2.0, 3.0), 63)])
def test_stokes_axis_aligned_mesh_vs_oracle(nc, upper, mask, stfem):
    """axis-aligned meshes (no vertex array: constant diagonal Jacobian path of the kernel) against the oracle: operator, vector mass and
    the space-time scatter"""
    from oracle import oracle
    nu = 0.7
    verts = stfem.mesh_vertices(nc, (0, 0, 0), upper)
    orc = oracle.StokesOracle(nc, verts, mask, nu)
    rng = np.random.default_rng(21)
    U, Pp = rng.uniform(-1, 1, 3 * orc.n_u), rng.uniform(-1, 1, orc.n_p)
    ku, kp = orc.apply(U, Pp)
    mu, _ = orc.apply(U, Pp, 0.0, 1.0)
    results = []
    for plane in ("-",):
        op = stfem.StokesMatrixFreeOperator(nc, upper=upper, dirichlet_mask=mask, viscosity=nu)
        u, p = op.initialize_dof_vector(0, U), op.initialize_dof_vector(1, Pp)
        ou, opr = op.initialize_dof_vector(0, np.full(3 * op.n_velocity, 7.0)), op.initialize_dof_vector(1, np.full(op.n_pressure, -3.0))
        op.vmult(ou, opr, u, p)
        m = op.initialize_dof_vector(0, np.full(3 * op.n_velocity, 5.0))
        op.mass_vmult(m, u)
        results.append((ou.download(), opr.download(), m.download()))
        assert rel(results[-1][0], ku.reshape(-1)) < TOL and rel(results[-1][2], mu.reshape(-1)) < TOL
        assert np.linalg.norm(results[-1][1] - kp) <= TOL * max(np.linalg.norm(kp), 1e-300) + 1e-14
        # space-time: cG(2), one step
        nt = 2
        Alpha, Beta, _, _ = stfem.get_fe_time_weights_stokes(stfem.CGP, 2, 1.0 / 16, 1)
        blocks = [None] * (2 * nt)
        r2 = np.random.default_rng(8)
        for d in range(nt):
            blocks[stfem.stokes_block_index(nt, 0, 0, d)] = r2.uniform(-1, 1, 3 * orc.n_u)
            blocks[stfem.stokes_block_index(nt, 0, 1, d)] = r2.uniform(-1, 1, orc.n_p)
        ref = orc.st_vmult(Alpha, Beta, 1, nt, blocks, True)
        var = [0 if b.size == 3 * orc.n_u else 1 for b in blocks]
        src = [op.initialize_dof_vector(v, b) for v, b in zip(var, blocks)]
        dst = [op.initialize_dof_vector(v, np.full(b.size, 11.0)) for v, b in zip(var, blocks)]
        op.st_vmult(Alpha, Beta, 1, nt, dst, src, True)
        for j in range(2 * nt):
            assert np.linalg.norm(dst[j].download() - ref[j]) <= TOL * np.linalg.norm(ref[j]) + 1e-14, (plane, j)


# ---- weak (Nitsche) boundary faces: StokesMatrixFreeOperator with weak_boundary_ids (LoopType::Full) and
# StokesNitscheMatrixFreeOperator (reference include/operators.h:1640-1741, 1768-1951)
NITSCHE_FIXTURES = ["stokes_nitsche_cart_2x2x2", "stokes_nitsche_pert_2x3x2", "stokes_nitsche_pert_3x2x2"]


def _ids(mask):
    return [f for f in range(6) if mask >> f & 1]


@pytest.mark.parametrize("name", NITSCHE_FIXTURES)
def test_stokes_nitsche_golden_fixture(name, stfem):
    """the dense numpy assembly of the face terms (tests/golden/make_golden_stokes.py: full 3D shape tables at the face points)"""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    op = stfem.StokesMatrixFreeOperator(tuple(g["ncell"]), vertices=g["vertices"], dirichlet_mask=int(g["mask"]), viscosity=float(g["nu"]),
                                        weak_boundary_ids=_ids(int(g["weak"])), penalty1=float(g["penalty1"]), penalty2=float(g["penalty2"]))
    u, p = op.initialize_dof_vector(0, g["U"]), op.initialize_dof_vector(1, g["P"])
    ou, opr = op.initialize_dof_vector(0, np.full(3 * op.n_velocity, 7.0)), op.initialize_dof_vector(1, np.full(op.n_pressure, -3.0))
    op.vmult(ou, opr, u, p)
    assert rel(ou.download(), g["SU"]) < TOL and rel(opr.download(), g["SP"]) < TOL
    # the right-hand-side functional of the Dirichlet data
    assert np.abs(op.face_points() - g["face_points"]).max() < 1e-14
    fu, fp = op.initialize_dof_vector(0), op.initialize_dof_vector(1)
    op.nitsche_rhs(g["G"], fu, fp)
    assert rel(fu.download(), g["FU"]) < TOL and rel(fp.download(), g["FP"]) < TOL
    op.nitsche_rhs(g["G"], fu, fp)  # accumulates (distribute_local_to_global)
    assert rel(fu.download(), 2 * g["FU"]) < TOL and rel(fp.download(), 2 * g["FP"]) < TOL


@pytest.mark.parametrize("nc,distort,mask,weak,outflow", [
    ((5, 4, 6), 0.15, 0, 0b111111, 0),             # every face weak: edge and corner cells carry two and three faces
    ((6, 5, 4), 0.1, 0b110000, 0b001011, 0b000100),  # strong z faces, weak x and upper-y faces, outflow at y = 0 (no term)
    ((3, 3, 3), 0.0, 0b000001, 0b101010, 0),       # axis-aligned cells given by vertices
])
def test_stokes_nitsche_vs_oracle(nc, distort, mask, weak, outflow, stfem):
    """operator with weak faces, space-time scatter (fused and unfused launches) and vmult_slice_add against the CPU oracle"""
    from oracle import oracle
    nu, pen1, pen2 = 0.3, 20.0, 10.0
    verts = stfem.mesh_vertices(nc, distort=distort, seed=41) if distort else stfem.mesh_vertices(nc)
    op = stfem.StokesMatrixFreeOperator(nc, vertices=verts, dirichlet_mask=mask, viscosity=nu, weak_boundary_ids=_ids(weak),
                                        outflow_boundary_ids=_ids(outflow), penalty1=pen1, penalty2=pen2)
    orc = oracle.StokesOracle(nc, verts, mask, nu, weak_mask=weak & ~outflow, penalty1=pen1, penalty2=pen2)
    rng = np.random.default_rng(9)
    U, Pp = rng.uniform(-1, 1, 3 * orc.n_u), rng.uniform(-1, 1, orc.n_p)
    ku, kp = orc.apply(U, Pp)
    u, p = op.initialize_dof_vector(0, U), op.initialize_dof_vector(1, Pp)
    ou, opr = op.initialize_dof_vector(0), op.initialize_dof_vector(1)
    op.vmult(ou, opr, u, p)
    assert rel(ou.download(), ku) < TOL and rel(opr.download(), kp) < TOL
    first = (ou.download(), opr.download())
    op.vmult(ou, opr, u, p)  # colour launches, plain read-add-write: reproducible to the bit
    assert np.array_equal(ou.download(), first[0]) and np.array_equal(opr.download(), first[1])
    # the vector mass has no face term
    mu, _ = orc.apply(U, Pp, 0.0, 1.0)
    m = op.initialize_dof_vector(0)
    op.mass_vmult(m, u)
    assert rel(m.download(), mu) < TOL
    for ns, r in ((1, 2), (2, 3)):  # 2 time dofs: one fused set of launches; 6: one launch set per source
        Alpha, Beta, _, _ = stfem.get_fe_time_weights_stokes(stfem.CGP, r, 1.0 / 16, ns)
        nt, nb = r, 2 * r * ns
        blocks = [None] * nb
        for it in range(ns):
            for d in range(nt):
                blocks[stfem.stokes_block_index(nt, it, 0, d)] = rng.uniform(-1, 1, 3 * orc.n_u)
                blocks[stfem.stokes_block_index(nt, it, 1, d)] = rng.uniform(-1, 1, orc.n_p)
        ref = orc.st_vmult(Alpha, Beta, ns, nt, blocks, True)
        var = [0 if b.size == 3 * orc.n_u else 1 for b in blocks]
        src = [op.initialize_dof_vector(v, b) for v, b in zip(var, blocks)]
        dst = [op.initialize_dof_vector(v, np.full(b.size, 11.0)) for v, b in zip(var, blocks)]
        op.st_vmult(Alpha, Beta, ns, nt, dst, src, True)
        for j in range(nb):
            assert rel(dst[j].download(), ref[j]) < TOL, (ns, r, j)
    # Dirichlet data functional against the oracle at the oracle's own face points
    pts = orc.face_points()
    assert np.abs(op.face_points() - pts).max() < 1e-13
    G = np.stack([np.sin(pts[:, 0] + 2 * pts[:, 1]), pts[:, 2] ** 2 - pts[:, 0], np.cos(pts[:, 1] * pts[:, 2])], axis=1)
    fu, fp = op.initialize_dof_vector(0), op.initialize_dof_vector(1)
    op.nitsche_rhs(G, fu, fp)
    ru, rp = orc.nitsche_rhs(G)
    assert rel(fu.download(), ru) < TOL and rel(fp.download(), rp) < TOL


def test_stokes_reference_Tvmult(stfem):
    """SystemMatrixStokes::Tvmult as the reference has it (operators.h:708-745: not a transpose, see the oracle's restatement)"""
    from oracle import oracle
    nc = (3, 2, 3)
    verts = stfem.mesh_vertices(nc, distort=0.1, seed=5)
    op = stfem.StokesMatrixFreeOperator(nc, vertices=verts, dirichlet_mask=63, viscosity=0.8)
    orc = oracle.StokesOracle(nc, verts, 63, 0.8)
    ns, r = 2, 2
    Alpha, Beta, _, _ = stfem.get_fe_time_weights_stokes(stfem.CGP, r, 1.0 / 16, ns)
    nt, nb = r, 2 * r * ns
    rng = np.random.default_rng(6)
    blocks = [None] * nb
    for it in range(ns):
        for d in range(nt):
            blocks[stfem.stokes_block_index(nt, it, 0, d)] = rng.uniform(-1, 1, 3 * orc.n_u)
            blocks[stfem.stokes_block_index(nt, it, 1, d)] = rng.uniform(-1, 1, orc.n_p)
    ref = orc.st_Tvmult(Alpha, Beta, ns, nt, blocks)
    var = [0 if b.size == 3 * orc.n_u else 1 for b in blocks]
    src = [op.initialize_dof_vector(v, b) for v, b in zip(var, blocks)]
    dst = [op.initialize_dof_vector(v, np.full(b.size, 3.0)) for v, b in zip(var, blocks)]
    op.st_Tvmult(Alpha, Beta, ns, nt, dst, src)
    for j in range(nb):
        assert np.linalg.norm(dst[j].download() - ref[j]) <= TOL * max(np.linalg.norm(ref[j]), 1.0), j
    assert any(np.linalg.norm(ref[j]) > 0 for j in range(nb))


# ---- FE_DGP(1) pressure: the reference's dGPressure = true (tests/tp_03stokes.cc:83-86, tests/json/stokes.json)
@pytest.mark.parametrize("name", ["stokes_dgp_cart_2x2x2", "stokes_dgp_pert_2x3x2"])
def test_stokes_dg_pressure_golden_fixture(name, stfem):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    weak = int(g["weak"])
    op = stfem.StokesMatrixFreeOperator(tuple(g["ncell"]), vertices=g["vertices"], dirichlet_mask=int(g["mask"]), viscosity=float(g["nu"]),
                                        weak_boundary_ids=_ids(weak), dg_pressure=True)
    assert op.n_pressure == 4 * int(np.prod(g["ncell"])) == g["P"].size
    u, p = op.initialize_dof_vector(0, g["U"]), op.initialize_dof_vector(1, g["P"])
    ou, opr = op.initialize_dof_vector(0, np.full(3 * op.n_velocity, 7.0)), op.initialize_dof_vector(1, np.full(op.n_pressure, -3.0))
    op.vmult(ou, opr, u, p)
    assert rel(ou.download(), g["SU"]) < TOL and rel(opr.download(), g["SP"]) < TOL
    mu = op.initialize_dof_vector(0)
    op.mass_vmult(mu, u)
    assert rel(mu.download(), g["MU"]) < TOL
    if weak:
        fu, fp = op.initialize_dof_vector(0), op.initialize_dof_vector(1)
        op.nitsche_rhs(g["G"], fu, fp)
        assert rel(fu.download(), g["FU"]) < TOL and rel(fp.download(), g["FP"]) < TOL


@pytest.mark.parametrize("nc,distort,mask,weak", [((5, 4, 6), 0.15, 63, 0), ((4, 3, 5), 0.1, 0b110000, 0b001111), ((6, 2, 3), 0.0, 0, 0b111111)])
def test_stokes_dg_pressure_vs_oracle(nc, distort, mask, weak, stfem):
    """Q2 / P1disc against the oracle: operator (with weak faces), space-time scatter (fused: 2 time dofs; unfused: 6)"""
    from oracle import oracle
    nu = 0.6
    verts = stfem.mesh_vertices(nc, distort=distort, seed=11) if distort else stfem.mesh_vertices(nc)
    op = stfem.StokesMatrixFreeOperator(nc, vertices=verts if distort else None, dirichlet_mask=mask, viscosity=nu, weak_boundary_ids=_ids(weak),
                                        dg_pressure=True)
    orc = oracle.StokesOracle(nc, verts, mask, nu, weak_mask=weak, dg_pressure=True)
    assert (op.n_velocity, op.n_pressure) == (orc.n_u, orc.n_p)
    rng = np.random.default_rng(2)
    U, Pp = rng.uniform(-1, 1, 3 * orc.n_u), rng.uniform(-1, 1, orc.n_p)
    ku, kp = orc.apply(U, Pp)
    ou, opr = op.initialize_dof_vector(0), op.initialize_dof_vector(1)
    op.vmult(ou, opr, op.initialize_dof_vector(0, U), op.initialize_dof_vector(1, Pp))
    assert rel(ou.download(), ku) < TOL and rel(opr.download(), kp) < TOL
    for ns, r in ((1, 2), (2, 3)):
        Alpha, Beta, _, _ = stfem.get_fe_time_weights_stokes(stfem.CGP, r, 1.0 / 16, ns)
        nt, nb = r, 2 * r * ns
        blocks = [None] * nb
        for it in range(ns):
            for d in range(nt):
                blocks[stfem.stokes_block_index(nt, it, 0, d)] = rng.uniform(-1, 1, 3 * orc.n_u)
                blocks[stfem.stokes_block_index(nt, it, 1, d)] = rng.uniform(-1, 1, orc.n_p)
        ref = orc.st_vmult(Alpha, Beta, ns, nt, blocks, True)
        var = [0 if b.size == 3 * orc.n_u else 1 for b in blocks]
        src = [op.initialize_dof_vector(v, b) for v, b in zip(var, blocks)]
        dst = [op.initialize_dof_vector(v, np.full(b.size, 11.0)) for v, b in zip(var, blocks)]
        op.st_vmult(Alpha, Beta, ns, nt, dst, src, True)
        for j in range(nb):
            assert rel(dst[j].download(), ref[j]) < TOL, (ns, r, j)


@pytest.mark.parametrize("world,dg,r", [(2, False, 2), (3, False, 2), (2, True, 2), (2, False, 1), (3, False, 1)])
def test_stokes_on_z_slabs_equals_whole_mesh(world, dg, r, stfem):
    """The Stokes operator shards like the scalar one (BASELINE configs[4] is an 8-GPU configuration): on a z-slab whose interface
    faces are taken out of the Dirichlet mask the kernels leave PARTIAL sums in the interface planes of the velocity components and of
    the FE_Q(1) pressure (the gather-form coupling kernels count the slab's own cells only; FE_DGP pressure DoFs are cell-local); adding
    the two sides' planes - the add-exchange stfem_halo_begin / end performs on the scalar views of the blocks - gives the whole-mesh
    result.  Here the exchange is done on the host."""
    nc, nu_ = (3, 2, 6), 0.8
    weak = [0]  # a weak (Nitsche) x- face: boundary cells of every slab
    mask = 63 & ~1
    # (r = 1: one time dof - the pressure gradient rides in the velocity sweep; r = 2: the separate gradient kernel)
    Alpha, Beta, _, _ = stfem.get_fe_time_weights_stokes(stfem.CGP, r, 0.1, 1)
    nt = r
    rng = np.random.default_rng(8)
    ndu = [2 * c + 1 for c in nc]
    ndp = [c + 1 for c in nc]
    whole = stfem.StokesMatrixFreeOperator(nc, dirichlet_mask=mask, viscosity=nu_, weak_boundary_ids=weak, dg_pressure=dg)
    var = [0] * nt + [1] * nt
    X = [rng.uniform(-1, 1, 3 * whole.n_velocity if v == 0 else whole.n_pressure) for v in var]
    src = [whole.initialize_dof_vector(v, x) for v, x in zip(var, X)]
    dst = [whole.initialize_dof_vector(v) for v in var]
    whole.st_vmult(Alpha, Beta, 1, nt, dst, src)
    want = [d.download() for d in dst]
    bounds = [round(nc[2] * r / world) for r in range(world + 1)]
    parts = []
    for r in range(world):
        z0, z1 = bounds[r], bounds[r + 1]
        m = mask
        if r > 0:
            m &= ~16
        if r < world - 1:
            m &= ~32
        snc = (nc[0], nc[1], z1 - z0)
        op = stfem.StokesMatrixFreeOperator(snc, lower=(0, 0, z0 / nc[2]), upper=(1, 1, z1 / nc[2]), dirichlet_mask=m, viscosity=nu_,
                                            weak_boundary_ids=weak, dg_pressure=dg)
        pu, pp = ndu[0] * ndu[1], ndp[0] * ndp[1]

        def cut(b, x):
            if var[b] == 0:
                return x.reshape(3, ndu[2], pu)[:, 2 * z0:2 * z1 + 1].reshape(-1)
            if dg:
                return x.reshape(nc[2], -1)[z0:z1].reshape(-1)
            return x.reshape(ndp[2], pp)[z0:z1 + 1].reshape(-1)

        s = [op.initialize_dof_vector(v, cut(b, X[b])) for b, v in enumerate(var)]
        d = [op.initialize_dof_vector(v) for v in var]
        op.st_vmult(Alpha, Beta, 1, nt, d, s)
        parts.append((z0, z1, [q.download() for q in d], cut))
    # the add-exchange: every interface plane is the sum of the two sides' partial sums
    for b, v in enumerate(var):
        if v == 1 and dg:
            got = np.concatenate([p[2][b] for p in parts])
            assert rel(got, want[b]) < TOL
            continue
        nz, plane, comps = (ndu[2], ndu[0] * ndu[1], 3) if v == 0 else (ndp[2], ndp[0] * ndp[1], 1)
        step = 2 if v == 0 else 1
        total = np.zeros((comps, nz, plane))
        for z0, z1, out, _ in parts:
            total[:, step * z0:step * z1 + 1] += out[b].reshape(comps, step * (z1 - z0) + 1, plane)
        assert rel(total.reshape(-1), want[b]) < TOL, b


def test_axpby_on_the_blocks_of_a_two_variable_vector(stfem):
    """stfem_axpby_many: the vector arithmetic of the Krylov solver and the multigrid on velocity and pressure blocks in one launch,
    with the zero-factor rules of stfem_vector_axpby (a zero factor means "not read": NaN does not survive an assignment)"""
    op = stfem.StokesMatrixFreeOperator((5, 4, 3))
    ctx = stfem.MatrixFreeOperator(2, (5, 4, 3))  # (any context of the device and precision)
    rng = np.random.default_rng(3)
    X = [rng.uniform(-1, 1, 3 * op.n_velocity), rng.uniform(-1, 1, op.n_pressure), rng.uniform(-1, 1, 3 * op.n_velocity)]
    Y = [rng.uniform(-1, 1, v.size) for v in X]
    var = [0, 1, 0]
    x = [op.initialize_dof_vector(v, h) for v, h in zip(var, X)]
    y = [op.initialize_dof_vector(v, h) for v, h in zip(var, Y)]
    stfem.axpby_many(ctx, 0.5, x, -2.0, y)
    for yv, xh, yh in zip(y, X, Y):
        assert np.array_equal(yv.download(), 0.5 * xh + -2.0 * yh)
    nan = [op.initialize_dof_vector(v, np.full(h.size, np.nan)) for v, h in zip(var, X)]
    stfem.axpby_many(ctx, 3.0, x, 0.0, nan)          # equ: the old content (NaN) is not read
    for nv, xh in zip(nan, X):
        assert np.array_equal(nv.download(), 3.0 * xh)
    bad = [op.initialize_dof_vector(v, np.full(h.size, np.inf)) for v, h in zip(var, X)]
    stfem.axpby_many(ctx, 0.0, bad, 0.0, bad)        # = 0
    for bv in bad:
        assert np.all(bv.download() == 0.0)
    stfem.axpby_many(ctx, 0.0, None, 2.0, y)         # scaling: x is not needed
    for yv, xh, yh in zip(y, X, Y):
        assert np.allclose(yv.download(), 2.0 * (0.5 * xh - 2.0 * yh), rtol=1e-15, atol=0)
