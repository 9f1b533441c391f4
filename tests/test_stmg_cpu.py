"""Space-time multigrid, CPU side: the level schedule of the product (host code behind the C-ABI) against the
known answers of the reference's own test tests/tp04.cc (data fixture tests/golden/mg_sequence_cases.json; tests/tp04.output
records that the reference passes them), the 1D factors of the product's space transfer against the cell-by-cell restatement
of deal.II's MGTwoLevelTransfer (oracle/stmg_oracle.py), and the properties that stand in for the missing reference numbers."""
import importlib
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp


@pytest.fixture(scope="module")
def stfem():
    return importlib.import_module("dealii-stfem_amd")


def test_mg_sequence_known_answers(stfem, golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "mg_sequence_cases.json")))
    assert len(cases) == 22
    for c in cases:
        # tests/tp04.cc passes an empty spatial degree sequence even with use_p_multigrid_space (it predates the parameter's use,
        # SURVEY 9.6); its expectations hold one p level per k level, i.e. a spatial sequence as long as the temporal one
        p_seq = [k + 1 for k in c["k_seq"]] if c["use_p_multigrid_space"] else []
        seq = stfem.get_mg_sequence(c["n_sp_lvl"], c["k_seq"], p_seq, c["n_timesteps_at_once"], c["n_timesteps_at_once_min"], c["lower_lvl"],
                                    c["coarsening_type"], c["time_before_space"], c["use_p_multigrid_space"], c["zip_from_back"])
        assert seq == c["expected_mg_type_level"], c["name"]
        if c["expected_precondition_types"] is not None:
            got = stfem.get_precondition_stmg_types(seq, c["coarsening_type"], c["time_before_space"])
            assert got == c["expected_precondition_types"], c["name"]


def test_poly_mg_sequence(stfem):
    # fe_time.cc:17-56
    assert stfem.get_poly_mg_sequence(4, 1, "bisect") == [1, 2, 4]
    assert stfem.get_poly_mg_sequence(5, 1, "bisect") == [1, 2, 5]
    assert stfem.get_poly_mg_sequence(4, 2, "decrease_by_one") == [2, 3, 4]
    assert stfem.get_poly_mg_sequence(4, 1, "go_to_one") == [1, 4]
    assert stfem.get_poly_mg_sequence(3, 3, "bisect") == [3]
    with pytest.raises(stfem.StfemError):
        stfem.get_poly_mg_sequence(1, 2)


def test_space_or_time_sequence(stfem):
    # fe_time.cc:94-101: one family after the other, finest-first families reversed
    assert stfem.get_mg_sequence(3, [1, 2], (), 2, 1, "t", "space_or_time", False) == "hhkt"
    assert stfem.get_mg_sequence(3, [1, 2], (), 2, 1, "t", "space_or_time", True) == "kthh"
    assert stfem.get_precondition_stmg_types("hhkt", "space_or_time") == [1] * 5


@pytest.mark.parametrize("pf,ncf,pc,ncc", [(2, (4, 2, 2), 2, (2, 1, 1)), (3, (2, 2, 2), 1, (2, 2, 2)), (4, (2, 2, 4), 2, (1, 1, 2)), (1, (4, 4, 2), 1, (2, 2, 2)),
                                           (2, (2, 4, 2), 2, (2, 2, 1))])
def test_line_factors_equal_cellwise_transfer(stfem, pf, ncf, pc, ncc):
    """Kronecker product of the product's 1D factors == the cell-by-cell assembly with inverse-valence weights"""
    from oracle import stmg_oracle
    lines = [stfem.transfer_line_matrices(ncf[d], pf, ncc[d], pc) for d in range(3)]
    P = sp.kron(lines[2][0], sp.kron(lines[1][0], lines[0][0])).toarray()
    want = stmg_oracle.space_prolongation(pf, ncf, 0, pc, ncc, 0).toarray()
    assert np.abs(P - want).max() < 1e-13
    I = sp.kron(lines[2][1], sp.kron(lines[1][1], lines[0][1])).toarray()
    assert np.abs(I - stmg_oracle.space_interpolation(pf, ncf, 0, pc, ncc, 0).toarray()).max() < 1e-13
    # the nodal interpolation is a left inverse of the embedding
    assert np.abs(I @ P - np.eye(P.shape[1])).max() < 1e-12


def test_transfer_properties(oracle_mod):
    """what stands in for reference numbers: the embedding reproduces the coarse space and the level operators are Galerkin"""
    from oracle import stmg_oracle
    pf, ncf, pc, ncc = 2, (4, 4, 2), 2, (2, 2, 1)
    P = stmg_oracle.space_prolongation(pf, ncf, 0, pc, ncc, 0)
    assert np.abs(P @ np.ones(P.shape[1]) - 1).max() < 1e-13
    # a function of the coarse space: a polynomial of degree pc in every variable
    def nodes(p, nc):
        g = np.asarray(oracle_mod.gauss_lobatto(p + 1))
        return [np.concatenate([(c + g[:-1]) / nc[d] for c in range(nc[d])] + [[1.0]]) for d in range(3)]
    f = lambda x, y, z: (1 + x - 2 * x * x) * (y * y + 0.5) * (3 - z + z * z)  # noqa: E731
    def sample(p, nc):
        x, y, z = nodes(p, nc)
        return f(x[None, None, :], y[None, :, None], z[:, None, None]).ravel()
    assert np.abs(P @ sample(pc, ncc) - sample(pf, ncf)).max() < 1e-12
    # Galerkin identity on nested spaces (Cartesian mesh): P^T M_f P = M_c, P^T K_f P = K_c, with and without constraints
    stfem = importlib.import_module("dealii-stfem_amd")
    for mask in (0, 63, 63 & ~3):
        Pm = stmg_oracle.space_prolongation(pf, ncf, mask, pc, ncc, mask).toarray()
        fine = oracle_mod.Oracle(pf, ncf, stfem.mesh_vertices(ncf), mask)
        coarse = oracle_mod.Oracle(pc, ncc, stfem.mesh_vertices(ncc), mask)
        for kw in ({"mass": 1.0}, {"laplace": 1.0}):
            assert np.abs(Pm.T @ fine.dense(**kw) @ Pm - coarse.dense(**kw)).max() < 1e-12
    # p-transfer
    Pp = stmg_oracle.space_prolongation(3, (2, 2, 2), 63, 1, (2, 2, 2), 63).toarray()
    fine = oracle_mod.Oracle(3, (2, 2, 2), stfem.mesh_vertices((2, 2, 2)), 63)
    coarse = oracle_mod.Oracle(1, (2, 2, 2), stfem.mesh_vertices((2, 2, 2)), 63)
    assert np.abs(Pp.T @ fine.dense(laplace=1.0) @ Pp - coarse.dense(laplace=1.0)).max() < 1e-12


def test_level_structure():
    from oracle import stmg_oracle
    # tests/tp04.cc-style sequence "ktth": temporal degree 2 -> 1 on the coarsest level, 4 -> 1 time steps
    got = stmg_oracle.level_structure(0, 4, "ktth", [1, 2])
    assert got == [(1, 1, 4.0), (2, 1, 4.0), (2, 2, 2.0), (2, 4, 1.0), (2, 4, 1.0)]


def test_vcycle_contracts(oracle_mod):
    """the restated V-cycle (h and tau levels, Vanka relaxation with the estimated parameter) is a contraction for the heat system"""
    from oracle import stmg_oracle, vanka_oracle
    stfem = importlib.import_module("dealii-stfem_amd")
    ttype, r, tau, n_steps, p = 1, 1, 0.125, 2, 1
    seq = "ht"  # coarsest transfer first: level 0 -(h)- level 1 -(tau)- level 2
    struct = stmg_oracle.level_structure(ttype, n_steps, seq, [r])
    meshes = [(2, 2, 2), (4, 4, 4), (4, 4, 4)]
    levels = []
    for l, (deg, n, scale) in enumerate(struct):
        nc = meshes[l]
        verts = stfem.mesh_vertices(nc)
        orc = oracle_mod.Oracle(p, nc, verts, 63)
        Alpha, Beta, _, _ = oracle_mod.time_weights(ttype, deg, tau * scale, n)
        A = np.kron(Alpha, orc.dense(laplace=1.0)) + np.kron(Beta, orc.dense(mass=1.0))
        van = vanka_oracle.VankaOracle(p, nc, verts, 63, Alpha, Beta)
        nb, N = Alpha.shape[0], A.shape[0] // Alpha.shape[0]
        sm = lambda v, van=van, nb=nb, N=N: van.vmult(v.reshape(nb, N)).ravel()  # noqa: E731
        levels.append(dict(A=A, smoother=sm, omega=stmg_oracle.power_iteration_relaxation(A, sm, nb, N), n_iterations=1))
    Ps = stmg_oracle.space_prolongation(p, meshes[1], 63, p, meshes[0], 63)
    nb1 = blk = stmg_oracle.blk_dofs(ttype, r) * struct[1][1]
    Pt, Rt = stmg_oracle.time_transfer(ttype, "t", r, r, struct[2][1])
    N2 = levels[2]["A"].shape[0] // Pt.shape[0]
    transfers = [None, (sp.kron(np.eye(nb1), Ps).tocsr(), sp.kron(np.eye(blk), Ps.T).tocsr()),
                 (sp.kron(Pt, sp.eye(N2)).tocsr(), sp.kron(Rt, sp.eye(N2)).tocsr())]
    mg = stmg_oracle.Multigrid(levels, transfers)
    A = levels[2]["A"]
    free = np.abs(A).sum(axis=1) > 0  # unconstrained rows
    rng = np.random.default_rng(3)
    x = np.where(free, rng.uniform(-1, 1, A.shape[0]), 0.0)
    b = A @ x
    u = np.zeros_like(b)
    res = [np.linalg.norm(b)]
    for _ in range(6):
        u = u + mg.vmult(b - A @ u)
        res.append(np.linalg.norm(b - A @ u))
    assert res[-1] < 1e-3 * res[0], res
