"""Geometric multigrid of the Stokes system (SURVEY 8 f-2 for BASELINE configs[4]; reference tests/tp_03stokes.cc:283-290, 484-770 and
include/stmg.h:1160-1419): one V-cycle of the C++ mirror (GMGStokes in host/stfem/stokes_solver.h: level operators, two-variable Vanka
relaxation smoothers, one space transfer per variable, all through the C-ABI) against the numpy V-cycle of oracle/stmg_oracle.py on
DENSE level matrices (Stokes oracle applied to unit vectors), the dense Vanka restatement and the cell-by-cell transfer restatement;
and the driver with the V-cycle as preconditioner."""
import os
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "dealii-stfem_amd", "host")


def _exe(name):
    exe = os.path.join(HOST, name)
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    return exe


def _dgp_prolongation(ncc):
    """FE_DGP(1) (deal.II's Legendre basis) of a mesh embedded into the mesh of its 2 x 2 x 2 children, cell by cell: with
    xi = (s + xi') / 2 the parent's l(xi) = sqrt 3 (2 xi - 1) is l(xi') / 2 + sqrt 3 (s - 1 / 2) on child s"""
    ncf = tuple(2 * c for c in ncc)
    rows, cols, vals = [], [], []
    s3h = np.sqrt(3.0) / 2
    for cz in range(ncf[2]):
        for cy in range(ncf[1]):
            for cx in range(ncf[0]):
                f = 4 * (cx + ncf[0] * (cy + ncf[1] * cz))
                c = 4 * ((cx // 2) + ncc[0] * ((cy // 2) + ncc[1] * (cz // 2)))
                off = [s3h if (q % 2) else -s3h for q in (cx, cy, cz)]
                rows += [f, f, f, f, f + 1, f + 2, f + 3]
                cols += [c, c + 1, c + 2, c + 3, c + 1, c + 2, c + 3]
                vals += [1.0, off[0], off[1], off[2], 0.5, 0.5, 0.5]
    nf, ncl = 4 * ncf[0] * ncf[1] * ncf[2], 4 * ncc[0] * ncc[1] * ncc[2]
    return sp.coo_matrix((vals, (rows, cols)), shape=(nf, ncl)).tocsr()


def _stokes_level(o, vanka_oracle, stfem, nc, ttype, r, tau, ns, nu, dg, omega, degree):
    """dense matrix and the smoother of one level; -> (dict for stmg_oracle.Multigrid, block sizes, BlockSlice order of the blocks)"""
    nt = r if ttype == 0 else r + 1
    Alpha, Beta, _, _ = stfem.get_fe_time_weights_stokes(ttype, r, tau, ns)
    var = [(b // nt) % 2 for b in range(2 * nt * ns)]  # BlockSlice(ns, 2, nt), variable-major: [step][variable][time dof]
    verts = stfem.mesh_vertices(nc)
    so = o.StokesOracle(nc, verts, 63, nu, dg_pressure=bool(dg))
    bs = [3 * so.n_u if v == 0 else so.n_p for v in var]
    off = np.concatenate([[0], np.cumsum(bs)])
    nb, N = len(bs), off[-1]
    A = np.zeros((N, N))
    e = [np.zeros(m) for m in bs]
    for b in range(nb):
        for j in range(bs[b]):
            e[b][j] = 1.0
            A[:, off[b] + j] = np.concatenate(so.st_vmult(Alpha, Beta, ns, nt, e))
            e[b][j] = 0.0
    vk = vanka_oracle.StokesVankaOracle(nc, verts, 63, nu, var, Alpha, Beta, dg_pressure=bool(dg))

    def smoother(rv):
        return np.concatenate(vk.vmult([rv[off[b]:off[b + 1]] for b in range(nb)]))

    return dict(A=A, smoother=smoother, omega=omega, n_iterations=degree), bs, (ns, nt)


def _time_transfer_matrix(Pt, fine, coarse, bs_f, bs_c):
    """Pt between the (step, time dof) blocks of ONE variable, placed on the blocks of both variables of the two BlockSlices"""
    (ns_f, nt_f), (ns_c, nt_c) = fine, coarse
    rows = [[None] * len(bs_c) for _ in bs_f]
    for i, m in enumerate(bs_f):
        for j, k in enumerate(bs_c):
            rows[i][j] = sp.csr_matrix((m, k))
    for v in range(2):
        for itf in range(ns_f):
            for idf in range(nt_f):
                for itc in range(ns_c):
                    for idc in range(nt_c):
                        w = Pt[itf * nt_f + idf, itc * nt_c + idc]
                        if w != 0.0:
                            i, j = itf * 2 * nt_f + v * nt_f + idf, itc * 2 * nt_c + v * nt_c + idc
                            rows[i][j] = w * sp.identity(bs_f[i], format="csr")
    return sp.bmat(rows, format="csr")


@pytest.mark.parametrize("n,levels,ttype,r,nu,degree,omega,variable,dg,seq,steps", [
    (4, 2, 0, 1, 1.0, 1, 0.6, 1, 0, None, 1),    # cG(1): one time dof; 4^3 -> 2^3 cells
    (4, 2, 1, 1, 0.5, 2, 0.5, 1, 0, None, 1),    # dG(1): two time dofs, two sweeps per smoothing step (a third level would be one cell: singular blocks)
    (4, 2, 0, 2, 1.0, 1, 0.6, 0, 0, None, 1),    # cG(2), one smoothing step on every level
    (4, 2, 0, 1, 1.0, 1, 0.4, 1, 1, None, 1),    # FE_DGP(1) pressure (the reference's default): the pressure transfer is the cell-wise embedding
    (4, 0, 0, 2, 1.0, 1, 0.5, 1, 0, "hk", 1),    # the reference's sequence for cG(2): coarsest 2^3 cells cG(1), h, then k to cG(2)
    (4, 0, 1, 1, 1.0, 1, 0.4, 1, 1, "kh", 1),    # dG(1) -> dG(0) on the coarse mesh last, FE_DGP(1) pressure
    (2, 0, 0, 1, 1.0, 1, 0.5, 1, 0, "t", 2),     # two time steps per slab -> one (tau level), same mesh
])
def test_stokes_vcycle_vs_oracle(n, levels, ttype, r, nu, degree, omega, variable, dg, seq, steps, tmp_path):
    from oracle import oracle as o, stmg_oracle as mg, vanka_oracle
    import importlib
    stfem = importlib.import_module("dealii-stfem_amd")
    out = str(tmp_path / "vc.bin")
    args = [_exe("test_host_stokes_mg"), str(n), str(levels), str(ttype), str(r), str(nu), str(degree), str(omega), str(variable), out, str(dg)]
    if seq is not None:
        args += [seq, str(steps)]
    res = subprocess.run(args, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    raw = np.fromfile(out, dtype=np.uint8)
    nb = int(np.frombuffer(raw[:8], dtype=np.uint64)[0])
    pos, X, sizes = 8, [], []
    for _ in range(nb):
        m = int(np.frombuffer(raw[pos:pos + 8], dtype=np.uint64)[0]); pos += 8
        X.append(np.frombuffer(raw[pos:pos + 8 * m], dtype=np.float64).copy()); pos += 8 * m
        sizes.append(m)
    Y = []
    for m in sizes:
        Y.append(np.frombuffer(raw[pos:pos + 8 * m], dtype=np.float64).copy()); pos += 8 * m
    nt = r if ttype == 0 else r + 1
    assert nb == 2 * nt * steps
    kinds = list(seq) if seq is not None else ["h"] * (levels - 1)  # transitions, coarse to fine
    # (cells, temporal degree, step size, steps per slab) of every level, finest last
    state, cfg = (n, r, 1.0 / 16, steps), []
    for kind in reversed(kinds):
        cfg.append(state)
        nl, rl, tau, ns = state
        state = (nl // 2, rl, tau, ns) if kind == "h" else ((nl, rl - 1, tau, ns) if kind == "k" else (nl, rl, 2 * tau, ns // 2))
    cfg.append(state)
    cfg.reverse()
    lv, transfers, shapes = [], [None], []
    for l, (nl, rl, tau, ns) in enumerate(cfg):
        level, bs, order = _stokes_level(o, vanka_oracle, stfem, (nl, nl, nl), ttype, rl, tau, ns, nu, dg, omega, degree)
        lv.append(level)
        shapes.append((bs, order))
        if l == 0:
            continue
        kind, (bs_c, order_c) = kinds[l - 1], shapes[l - 1]
        if kind == "h":
            nc, ncc = (nl, nl, nl), (nl // 2,) * 3
            Pu = mg.space_prolongation(2, nc, 63, 2, ncc, 63)
            Pp = _dgp_prolongation(ncc) if dg else mg.space_prolongation(1, nc, 0, 1, ncc, 0)
            nt_l = order[1]
            P = sp.block_diag([sp.block_diag([Pu, Pu, Pu]) if (b // nt_l) % 2 == 0 else Pp for b in range(len(bs))]).tocsr()
        else:
            Pt, _ = mg.time_transfer(ttype, kind, rl, cfg[l - 1][1], ns)
            P = _time_transfer_matrix(Pt, order, order_c, bs, bs_c)
        transfers.append((P, P.T.tocsr()))
    assert [len(x) for x in X] == shapes[-1][0]
    want = mg.Multigrid(lv, transfers, variable=bool(variable), steps=1).vmult(np.concatenate(X))
    got = np.concatenate(Y)
    rel = np.linalg.norm(got - want) / np.linalg.norm(want)
    assert rel < 1e-9, rel


def test_stokes_driver_with_multigrid():
    """The same rows whatever the preconditioner; the V-cycle needs far fewer FGMRES iterations than the smoother alone, and about
    as many on the finer mesh."""
    from oracle import slab_oracle
    exe = _exe("stokes_convergence")
    rows = {}
    for refinement, extra in ((2, []), (2, ["mg=2"]), (3, ["mg=3"])):
        # two sweeps per smoothing step, relaxation estimated per level (0), two slabs
        res = subprocess.run([exe, "0", "1", str(refinement), "2", "0", "1.0", str(2 ** refinement), "0.125"] + extra, capture_output=True, text=True,
                             timeout=900)
        assert res.returncode == 0, res.stdout + res.stderr
        rows[(refinement, bool(extra))] = [float(v) for v in res.stdout.split()]
    plain, mg2, mg3 = rows[(2, False)], rows[(2, True)], rows[(3, True)]
    assert np.allclose(plain[4:8], mg2[4:8], rtol=1e-7, atol=1e-10)
    assert mg2[8] < 0.8 * plain[8], (plain[8], mg2[8])
    assert mg3[8] < 1.5 * mg2[8] + 2, (mg2[8], mg3[8])


def test_stokes_driver_with_space_time_levels():
    """cG(2): the reference's level sequence (h levels, then the k level from cG(1) to cG(2)) as preconditioner gives the rows of the
    space-only multigrid and of the smoother alone (the solution does not depend on the preconditioner)"""
    exe = _exe("stokes_convergence")
    rows, seqs = [], []
    for extra in ([], ["mg=2"], ["mg=2", "stmg=1"], ["mg=2", "stmg=1", "coarsening=space_and_time"]):
        res = subprocess.run([exe, "0", "2", "2", "2", "0", "1.0", "4", "0.125"] + extra, capture_output=True, text=True, timeout=900)
        assert res.returncode == 0, res.stdout + res.stderr
        rows.append([float(v) for v in res.stdout.split()])
        seqs.append([ln.split(":")[1].split() for ln in res.stderr.splitlines() if ln.startswith("levels:")])
    assert seqs[2] == [["h", "k"]] and seqs[3] in ([["h", "k"]], [["k", "h"]])
    for r in rows[1:]:
        assert np.allclose(rows[0][4:8], r[4:8], rtol=1e-7, atol=1e-10)
        assert r[8] < rows[0][8]
