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


@pytest.mark.parametrize("n,levels,ttype,r,nu,degree,omega,variable,dg", [
    (4, 2, 0, 1, 1.0, 1, 0.6, 1, 0),    # cG(1): one time dof; 4^3 -> 2^3 cells
    (4, 2, 1, 1, 0.5, 2, 0.5, 1, 0),    # dG(1): two time dofs, two sweeps per smoothing step (a third level would be one cell: singular blocks)
    (4, 2, 0, 2, 1.0, 1, 0.6, 0, 0),    # cG(2), one smoothing step on every level
    (4, 2, 0, 1, 1.0, 1, 0.4, 1, 1),    # FE_DGP(1) pressure (the reference's default): the pressure transfer is the cell-wise embedding
])
def test_stokes_vcycle_vs_oracle(n, levels, ttype, r, nu, degree, omega, variable, dg, tmp_path):
    from oracle import oracle as o, stmg_oracle as mg, vanka_oracle
    import importlib
    stfem = importlib.import_module("dealii-stfem_amd")
    out = str(tmp_path / "vc.bin")
    res = subprocess.run([_exe("test_host_stokes_mg"), str(n), str(levels), str(ttype), str(r), str(nu), str(degree), str(omega), str(variable), out, str(dg)],
                         capture_output=True, text=True, timeout=600)
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
    assert nb == 2 * nt
    Alpha, Beta, _, _ = stfem.get_fe_time_weights_stokes(ttype, r, 1.0 / 16, 1)
    var = [0] * nt + [1] * nt  # BlockSlice(1, 2, nt), variable-major
    lv, transfers = [], [None]
    for l in range(levels):
        nl = n >> (levels - 1 - l)
        nc = (nl, nl, nl)
        verts = stfem.mesh_vertices(nc)
        so = o.StokesOracle(nc, verts, 63, nu, dg_pressure=bool(dg))
        Nu, Np = so.n_u, so.n_p
        bs = [3 * Nu] * nt + [Np] * nt
        off = np.concatenate([[0], np.cumsum(bs)])
        N = off[-1]
        A = np.zeros((N, N))
        e = [np.zeros(s) for s in bs]
        for b in range(nb):
            for j in range(bs[b]):
                e[b][j] = 1.0
                col = so.st_vmult(Alpha, Beta, 1, nt, e)
                A[:, off[b] + j] = np.concatenate(col)
                e[b][j] = 0.0
        vk = vanka_oracle.StokesVankaOracle(nc, verts, 63, nu, var, Alpha, Beta, dg_pressure=bool(dg))

        def smoother(rv, vk=vk, off=off):
            return np.concatenate(vk.vmult([rv[off[b]:off[b + 1]] for b in range(nb)]))

        lv.append(dict(A=A, smoother=smoother, omega=omega, n_iterations=degree))
        if l > 0:
            ncc = (nl // 2,) * 3
            Pu = mg.space_prolongation(2, nc, 63, 2, ncc, 63)
            Pp = _dgp_prolongation(ncc) if dg else mg.space_prolongation(1, nc, 0, 1, ncc, 0)
            blocks = [sp.block_diag([Pu, Pu, Pu]) if v == 0 else Pp for v in var]
            P = sp.block_diag(blocks).tocsr()
            transfers.append((P, P.T.tocsr()))
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
