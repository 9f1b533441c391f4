#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ from FIRST PRINCIPLES.

Independent of oracle/stfem_oracle.c and of the HIP library:
  * temporal matrices: 60-digit mpmath/sympy, exact polynomial integration (no quadrature),
  * spatial K, M: dense numpy assembly with full 3D shape-function tables (no sum
    factorisation), numpy's own Gauss/Lobatto nodes,
  * space-time vmult: dense Kronecker formula  dst_j = sum_i Alpha(j,i) K src_i + Beta(j,i) M src_i
    (reference: include/operators.h:536-559).

The reference itself (deal.II based) cannot be built or run in this image, so no fixture here
is an output of the reference; the reference's own golden file for the temporal matrices
(tests/tp_02.output, a data file) is committed next to these as tp_02.output and compared in
tests/test_time_weights.py.

Run:  python tests/golden/make_golden.py     (rewrites *.npz / *.json in this directory)
"""
import json
import os

import mpmath as mp
import numpy as np
import sympy as sp
from numpy.polynomial import legendre as L

HERE = os.path.dirname(os.path.abspath(__file__))
mp.mp.dps = 60

# ----------------------------------------------------------------------------- temporal matrices


def _legendre_roots(poly_expr, x):
    return sorted(sp.Poly(poly_expr, x).nroots(n=60, maxsteps=200), key=lambda r: sp.re(r))


def gll01(n):
    x = sp.symbols("x")
    pts = [sp.Integer(-1), sp.Integer(1)]
    if n > 2:
        pts += [sp.re(r) for r in _legendre_roots(sp.diff(sp.legendre(n - 1, x), x), x)]
    return sorted([(sp.Float(p, 60) + 1) / 2 for p in pts])


def radau_right01(n):
    x = sp.symbols("x")
    expr = sp.cancel((sp.legendre(n, x) - sp.legendre(n - 1, x)) / (x - 1))
    pts = [sp.Integer(1)]
    if n > 1:
        pts += [sp.re(r) for r in _legendre_roots(expr, x)]
    return sorted([(sp.Float(p, 60) + 1) / 2 for p in pts])


def lagrange_polys(pts):
    x = sp.symbols("x")
    polys = []
    for a, xa in enumerate(pts):
        e = sp.Integer(1)
        for b, xb in enumerate(pts):
            if b != a:
                e = e * (x - xb) / (xa - xb)
        polys.append(sp.Poly(sp.expand(e), x))
    return polys


def _int01(poly):
    x = poly.gens[0]
    F = poly.integrate(x)
    return F.eval(1) - F.eval(0)


def cg_weights(r):
    """fe_time.h:643-696: trial = Lagrange on GLL(r+1), test = Lagrange on those minus the first."""
    trial_pts = gll01(r + 1)
    trial = lagrange_polys(trial_pts)
    test = lagrange_polys(trial_pts[1:])
    M = [[_int01(test[i] * trial[j]) for j in range(r + 1)] for i in range(r)]
    D = [[_int01(test[i] * trial[j].diff()) for j in range(r + 1)] for i in range(r)]
    return np.array(M, dtype=float), np.array(D, dtype=float)


def dg_weights(r):
    """fe_time.h:698-744: Lagrange on right Radau(r+1); jump term phi_i(0) phi_j(0)."""
    polys = lagrange_polys(radau_right01(r + 1))
    jump = [p.eval(0) for p in polys]
    M = [[_int01(polys[i] * polys[j]) for j in range(r + 1)] for i in range(r + 1)]
    D = [[jump[i] * jump[j] + _int01(polys[i] * polys[j].diff()) for j in range(r + 1)]
         for i in range(r + 1)]
    return np.array(M, dtype=float), np.array(D, dtype=float), np.array(jump, dtype=float)


# ----------------------------------------------------------------------------- spatial operators


def gauss01(n):
    x, w = L.leggauss(n)
    return (x + 1) / 2, w / 2


def gll01_np(n):
    if n == 2:
        return np.array([0.0, 1.0])
    inner = L.Legendre.basis(n - 1).deriv().roots()
    return (np.concatenate([[-1.0], np.sort(inner.real), [1.0]]) + 1) / 2


def lagrange_table(nodes, x):
    """V[q,a] = l_a(x_q), G[q,a] = l_a'(x_q) through np.poly1d (independent of the C oracle)."""
    n = len(nodes)
    V = np.zeros((len(x), n))
    G = np.zeros((len(x), n))
    for a in range(n):
        others = np.delete(nodes, a)
        poly = np.poly1d(others, r=True) / np.prod(nodes[a] - others)
        V[:, a] = poly(x)
        G[:, a] = poly.deriv()(x)
    return V, G


def structured_vertices(ncell, lower, upper, jitter=0.0, seed=0):
    nv = [n + 1 for n in ncell]
    gx = [np.linspace(lower[d], upper[d], nv[d]) for d in range(3)]
    Z, Y, X = np.meshgrid(gx[2], gx[1], gx[0], indexing="ij")
    v = np.stack([X, Y, Z], axis=-1)  # [k][j][i][xyz], x fastest
    if jitter:
        rng = np.random.default_rng(seed)
        h = [(upper[d] - lower[d]) / ncell[d] for d in range(3)]
        d = rng.uniform(-1, 1, size=v.shape) * jitter * np.array(h)
        interior = np.zeros(v.shape[:3], dtype=bool)
        interior[1:-1, 1:-1, 1:-1] = True
        v = v + d * interior[..., None]
    return v.reshape(-1, 3).copy()


def dense_operators(p, ncell, vertices, dirichlet_mask, coef_lap=None, coef_mass=None):
    """Dense constrained K (stiffness) and M (mass) by direct quadrature."""
    n1 = p + 1
    nq = p + 1
    xq, wq = gauss01(nq)
    V, G = lagrange_table(gll01_np(n1), xq)
    # 3D tables [qz,qy,qx, c,b,a]
    N = np.einsum("zc,yb,xa->zyxcba", V, V, V).reshape(nq ** 3, n1 ** 3)
    dN = np.stack([
        np.einsum("zc,yb,xa->zyxcba", V, V, G).reshape(nq ** 3, n1 ** 3),
        np.einsum("zc,yb,xa->zyxcba", V, G, V).reshape(nq ** 3, n1 ** 3),
        np.einsum("zc,yb,xa->zyxcba", G, V, V).reshape(nq ** 3, n1 ** 3)], axis=1)  # [q,e,a]
    W = np.einsum("z,y,x->zyx", wq, wq, wq).reshape(-1)
    QX = np.tile(xq, nq * nq)
    QY = np.tile(np.repeat(xq, nq), nq)
    QZ = np.repeat(xq, nq * nq)
    nd = [p * n + 1 for n in ncell]
    ndofs = nd[0] * nd[1] * nd[2]
    K = np.zeros((ndofs, ndofs))
    M = np.zeros((ndofs, ndofs))
    nvx, nvy = ncell[0] + 1, ncell[1] + 1
    verts = vertices.reshape(-1, 3)
    for cz in range(ncell[2]):
        for cy in range(ncell[1]):
            for cx in range(ncell[0]):
                cell = cx + ncell[0] * (cy + ncell[1] * cz)
                X = np.array([verts[(cx + i) + nvx * ((cy + j) + nvy * (cz + k))]
                              for k in range(2) for j in range(2) for i in range(2)])  # [v,d]
                fx = np.stack([1 - QX, QX], 1); fy = np.stack([1 - QY, QY], 1)
                fz = np.stack([1 - QZ, QZ], 1)
                dd = np.array([-1.0, 1.0])
                J = np.zeros((nq ** 3, 3, 3))
                for k in range(2):
                    for j in range(2):
                        for i in range(2):
                            Xv = X[i + 2 * j + 4 * k]
                            J[:, :, 0] += np.outer(dd[i] * fy[:, j] * fz[:, k], Xv)
                            J[:, :, 1] += np.outer(fx[:, i] * dd[j] * fz[:, k], Xv)
                            J[:, :, 2] += np.outer(fx[:, i] * fy[:, j] * dd[k], Xv)
                det = np.linalg.det(J)
                Jinv = np.linalg.inv(J)  # [q,e,d] = d xi_e / d x_d
                gphys = np.einsum("qed,qea->qda", Jinv, dN)
                cl = np.ones(nq ** 3) if coef_lap is None else coef_lap[cell]
                cm = np.ones(nq ** 3) if coef_mass is None else coef_mass[cell]
                Ke = np.einsum("q,qda,qdb->ab", W * det * cl, gphys, gphys)
                Me = np.einsum("q,qa,qb->ab", W * det * cm, N, N)
                idx = np.array([(p * cx + a) + nd[0] * ((p * cy + b) + nd[1] * (p * cz + c))
                                for c in range(n1) for b in range(n1) for a in range(n1)])
                K[np.ix_(idx, idx)] += Ke
                M[np.ix_(idx, idx)] += Me
    con = np.zeros(nd[::-1], dtype=bool)  # [k,j,i]
    if dirichlet_mask & 1: con[:, :, 0] = True
    if dirichlet_mask & 2: con[:, :, -1] = True
    if dirichlet_mask & 4: con[:, 0, :] = True
    if dirichlet_mask & 8: con[:, -1, :] = True
    if dirichlet_mask & 16: con[0, :, :] = True
    if dirichlet_mask & 32: con[-1, :, :] = True
    con = con.reshape(-1)
    # homogeneous constraints: constrained src entries read as 0, constrained rows never written
    K[con, :] = 0; K[:, con] = 0; M[con, :] = 0; M[:, con] = 0
    return K, M


def time_weights_single(kind, r, tau):
    """Alpha, Beta of a single step (fe_time.h:351-372 + split_lhs_rhs 485-514)."""
    if kind == "cg":
        Mt, Dt = cg_weights(r)
        return tau * Mt[:, 1:], Dt[:, 1:]
    Mt, Dt, _ = dg_weights(r)
    return tau * Mt, Dt


def main():
    # ---- temporal matrices
    tw = {}
    for r in range(1, 6):
        M, D = cg_weights(r)
        tw[f"cg{r}"] = {"M": M.tolist(), "D": D.tolist()}
    for r in range(0, 6):
        M, D, j = dg_weights(r)
        tw[f"dg{r}"] = {"M": M.tolist(), "D": D.tolist(), "jump": j.tolist()}
    with open(os.path.join(HERE, "time_weights_exact.json"), "w") as f:
        json.dump(tw, f, indent=1)

    # ---- spatial + space-time fixtures
    cases = [
        # name, p, ncell, lower, upper, jitter, mask, time kind, r, tau, store_dense
        ("q1_cart_3x3x3", 1, (3, 3, 3), (0, 0, 0), (1, 1, 1), 0.0, 63, "cg", 1, 0.25, True),
        ("q2_cart_2x2x2", 2, (2, 2, 2), (0, 0, 0), (1, 1.5, 0.7), 0.0, 63, "cg", 1, 1 / 32, True),
        ("q2_pert_2x3x2", 2, (2, 3, 2), (0, 0, 0), (1, 1, 1), 0.15, 63, "cg", 2, 0.1, True),
        ("q2_free_2x2x2", 2, (2, 2, 2), (-1, -1, -1), (1, 1, 1), 0.0, 0, "dg", 1, 0.5, True),
        ("q3_pert_2x2x2", 3, (2, 2, 2), (-1, -1, -1), (1, 1, 1), 0.1, 63, "dg", 2, 1 / 64, False),
        ("q4_cart_2x2x2", 4, (2, 2, 2), (0, 0, 0), (1, 1, 1), 0.0, 63, "cg", 2, 1 / 144, False),
        ("q4_pert_3x2x2", 4, (3, 2, 2), (0, 0, 0), (1, 1, 1), 0.15, 0b010101, "cg", 2, 1 / 288,
         False),
    ]
    for (name, p, ncell, lo, up, jit, mask, kind, r, tau, dense) in cases:
        rng = np.random.default_rng(abs(hash(name)) % (2 ** 31) if False else sum(map(ord, name)))
        verts = structured_vertices(ncell, lo, up, jit, seed=17)
        ncells = ncell[0] * ncell[1] * ncell[2]
        coef = None
        if "pert" in name:  # variable (per-quadrature-point) laplace coefficient
            coef = rng.uniform(0.5, 2.0, size=(ncells, (p + 1) ** 3))
        K, M = dense_operators(p, ncell, verts, mask, coef_lap=coef)
        n = K.shape[0]
        Alpha, Beta = time_weights_single(kind, r, tau)
        nb = Alpha.shape[0]
        X = rng.uniform(-1, 1, size=(nb, n))
        Y = Alpha @ (X @ K.T) + Beta @ (X @ M.T)          # vmult
        YT = Alpha.T @ (X @ K.T) + Beta.T @ (X @ M.T)     # Tvmult
        out = dict(p=p, ncell=np.array(ncell), vertices=verts, mask=mask, Alpha=Alpha, Beta=Beta,
                   X=X, Y=Y, YT=YT, KX=X @ K.T, MX=X @ M.T, diagK=np.diag(K).copy(),
                   diagM=np.diag(M).copy())
        if coef is not None:
            out["coef_lap"] = coef
        if dense:
            out["K"] = K
            out["M"] = M
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "ndofs", n, "nb", nb, "|K|", np.abs(K).max(), "sym",
              np.abs(K - K.T).max(), np.abs(M - M.T).max())


if __name__ == "__main__":
    main()
