#!/usr/bin/env python3
"""Golden fixtures of the Stokes two-field operator, from FIRST PRINCIPLES:
dense numpy assembly with full 3D shape-function tables, independent of oracle/*.c and of the
HIP library.

  spatial:    out_u = nu K u - B^T p,  out_p = B u           (reference include/operators.h:1525-1575:
              pressure.submit_value(div u); velocity.submit_gradient(nu grad u - p I))
  space-time: SystemMatrixStokes::vmult (operators.h:696-700, 825-867): for every source time dof i
              dst[j, v] += Alpha(j_v, i_0) * (K_S (u_i, p_i))_v ;  dst[j, 0] += Beta(j_0, i_0) * M u_i
              with Alpha, Beta laid out as get_fe_time_weights_stokes does (fe_time.h:1242-1285)
              and blocks numbered by BlockSlice (fe_time.h:956-967), variable-major.

Velocity FE_Q(2)^3, pressure FE_Q(1), QGauss(3), MappingQ1; homogeneous Dirichlet on the velocity.
Layout: velocity block = 3 component arrays (component-major) of the scalar Q2 numbering.

Run:  python tests/golden/make_golden_stokes.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import (gauss01, gll01_np, lagrange_table, structured_vertices,  # noqa: E402
                         time_weights_single)


def tables3d(p, xq):
    V, G = lagrange_table(gll01_np(p + 1), xq)
    nq, n1 = len(xq), p + 1
    N = np.einsum("zc,yb,xa->zyxcba", V, V, V).reshape(nq ** 3, n1 ** 3)
    dN = np.stack([
        np.einsum("zc,yb,xa->zyxcba", V, V, G).reshape(nq ** 3, n1 ** 3),
        np.einsum("zc,yb,xa->zyxcba", V, G, V).reshape(nq ** 3, n1 ** 3),
        np.einsum("zc,yb,xa->zyxcba", G, V, V).reshape(nq ** 3, n1 ** 3)], axis=1)
    return N, dN


def dgp_values(pp, X, Y, Z):
    """FE_DGP(pp) on the reference cell as deal.II builds it (PolynomialSpace of the orthonormal Legendre polynomials on [0, 1],
    complete degree pp, x index fastest): values [point, basis function]"""
    def leg(n, x):
        t = 2 * x - 1
        return np.sqrt(2 * n + 1) * np.polynomial.legendre.legval(t, [0] * n + [1])
    cols = []
    for k in range(pp + 1):
        for j in range(pp + 1 - k):
            for i in range(pp + 1 - k - j):
                cols.append(leg(i, X) * leg(j, Y) * leg(k, Z))
    return np.stack(cols, axis=1)


def dense_stokes(pu, ncell, vertices, mask, dgp=False):
    pp = pu - 1
    nq = pu + 1
    xq, wq = gauss01(nq)
    Nu_, dNu = tables3d(pu, xq)
    Np_, _ = tables3d(pp, xq)
    W = np.einsum("z,y,x->zyx", wq, wq, wq).reshape(-1)
    QX = np.tile(xq, nq * nq); QY = np.tile(np.repeat(xq, nq), nq); QZ = np.repeat(xq, nq * nq)
    ndu = [pu * n + 1 for n in ncell]
    ndp = [pp * n + 1 for n in ncell]
    NU, NP = int(np.prod(ndu)), int(np.prod(ndp))
    if dgp:  # discontinuous pressure: the cell's own functions
        Np_ = dgp_values(pp, QX, QY, QZ)
        NP = int(np.prod(ncell)) * Np_.shape[1]
    K = np.zeros((NU, NU)); M = np.zeros((NU, NU)); B = np.zeros((3, NP, NU))
    nvx, nvy = ncell[0] + 1, ncell[1] + 1
    verts = vertices.reshape(-1, 3)
    for cz in range(ncell[2]):
        for cy in range(ncell[1]):
            for cx in range(ncell[0]):
                X = np.array([verts[(cx + i) + nvx * ((cy + j) + nvy * (cz + k))]
                              for k in range(2) for j in range(2) for i in range(2)])
                fx = np.stack([1 - QX, QX], 1); fy = np.stack([1 - QY, QY], 1); fz = np.stack([1 - QZ, QZ], 1)
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
                Jinv = np.linalg.inv(J)
                g = np.einsum("qed,qea->qda", Jinv, dNu)
                Ke = np.einsum("q,qda,qdb->ab", W * det, g, g)
                Me = np.einsum("q,qa,qb->ab", W * det, Nu_, Nu_)
                Be = np.einsum("q,qb,qda->dba", W * det, Np_, g)
                iu = np.array([(pu * cx + a) + ndu[0] * ((pu * cy + b) + ndu[1] * (pu * cz + c))
                               for c in range(pu + 1) for b in range(pu + 1) for a in range(pu + 1)])
                ip = np.array([(pp * cx + a) + ndp[0] * ((pp * cy + b) + ndp[1] * (pp * cz + c))
                               for c in range(pp + 1) for b in range(pp + 1) for a in range(pp + 1)])
                if dgp:
                    ip = (cx + ncell[0] * (cy + ncell[1] * cz)) * Np_.shape[1] + np.arange(Np_.shape[1])
                K[np.ix_(iu, iu)] += Ke
                M[np.ix_(iu, iu)] += Me
                for d in range(3):
                    B[d][np.ix_(ip, iu)] += Be[d]
    con = np.zeros(ndu[::-1], dtype=bool)
    if mask & 1: con[:, :, 0] = True
    if mask & 2: con[:, :, -1] = True
    if mask & 4: con[:, 0, :] = True
    if mask & 8: con[:, -1, :] = True
    if mask & 16: con[0, :, :] = True
    if mask & 32: con[-1, :, :] = True
    con = con.reshape(-1)
    K[con, :] = 0; K[:, con] = 0; M[con, :] = 0; M[:, con] = 0
    B[:, :, con] = 0
    return K, M, B


def face_tables(p, pts):
    """values / reference gradients of the 3D Lagrange basis at the tensor points pts[2] x pts[1] x pts[0] (x fastest)"""
    nodes = gll01_np(p + 1)
    VG = [lagrange_table(nodes, np.asarray(pt, dtype=float)) for pt in pts]
    (Vx, Gx), (Vy, Gy), (Vz, Gz) = VG
    n1 = p + 1
    nq = len(pts[0]) * len(pts[1]) * len(pts[2])
    N = np.einsum("zc,yb,xa->zyxcba", Vz, Vy, Vx).reshape(nq, n1 ** 3)
    dN = np.stack([np.einsum("zc,yb,xa->zyxcba", Vz, Vy, Gx).reshape(nq, n1 ** 3),
                   np.einsum("zc,yb,xa->zyxcba", Vz, Gy, Vx).reshape(nq, n1 ** 3),
                   np.einsum("zc,yb,xa->zyxcba", Gz, Vy, Vx).reshape(nq, n1 ** 3)], axis=1)
    return N, dN


def dense_nitsche(pu, ncell, vertices, mask, weak, nu, penalty1, penalty2, gfun, dgp=False):
    """Weak (Nitsche) boundary faces of the linear Stokes operator, reference include/operators.h:1713-1741:
         a_F((u,p),(v,q)) = int_F  -nu (grad u n).v + p n.v + gamma1/h u.v + gamma2/h (u.n)(v.n) - nu u.(grad v n) - q u.n
       with gamma1 = nu penalty1, gamma2 = penalty2, h = sqrt(area of the face)  (operators.h:184-209, 1220-1221)
       and the functional of StokesNitscheMatrixFreeOperator (operators.h:1898-1940) for the Dirichlet data gfun(x).
       Returns Auu [3 NU, 3 NU], Aup [3 NU, NP], Apu [NP, 3 NU], Fu [3 NU], Fp [NP], face points, g at them."""
    pp, nq = pu - 1, pu + 1
    xq, wq = gauss01(nq)
    ndu = [pu * n + 1 for n in ncell]; ndp = [pp * n + 1 for n in ncell]
    NU, NP = int(np.prod(ndu)), int(np.prod(ndp))
    if dgp:
        NP = int(np.prod(ncell)) * ((pp + 1) * (pp + 2) * (pp + 3) // 6)
    Auu = np.zeros((3 * NU, 3 * NU)); Aup = np.zeros((3 * NU, NP)); Apu = np.zeros((NP, 3 * NU))
    Fu = np.zeros(3 * NU); Fp = np.zeros(NP)
    g1, g2 = nu * penalty1, penalty2
    nvx, nvy = ncell[0] + 1, ncell[1] + 1
    verts = vertices.reshape(-1, 3)
    all_pts, all_g = [], []
    for f in range(6):
        if not weak & (1 << f):
            continue
        d, sd = f // 2, f % 2
        t1, t2 = [e for e in range(3) if e != d]
        pts = [None] * 3
        pts[d] = [float(sd)]; pts[t1] = xq; pts[t2] = xq
        Nu_, dNu = face_tables(pu, pts)
        Np_, _ = face_tables(pp, pts)
        Wf = np.einsum("b,a->ba", wq, wq).reshape(-1)  # q = q1 + nq q2 (the point tables run x fastest: t1 < t2)
        grid = np.meshgrid(*[np.asarray(pts[2]), np.asarray(pts[1]), np.asarray(pts[0])], indexing="ij")
        QZ, QY, QX = [gq.reshape(-1) for gq in grid]
        if dgp:
            Np_ = dgp_values(pp, QX, QY, QZ)
        for c2 in range(ncell[t2]):
            for c1 in range(ncell[t1]):
                cc = [0, 0, 0]
                cc[d] = ncell[d] - 1 if sd else 0; cc[t1] = c1; cc[t2] = c2
                cx, cy, cz = cc
                X = np.array([verts[(cx + i) + nvx * ((cy + j) + nvy * (cz + k))] for k in range(2) for j in range(2) for i in range(2)])
                fx = np.stack([1 - QX, QX], 1); fy = np.stack([1 - QY, QY], 1); fz = np.stack([1 - QZ, QZ], 1)
                dd = np.array([-1.0, 1.0])
                nqf = len(QX)
                J = np.zeros((nqf, 3, 3)); xyz = np.zeros((nqf, 3))
                for k in range(2):
                    for j in range(2):
                        for i in range(2):
                            Xv = X[i + 2 * j + 4 * k]
                            J[:, :, 0] += np.outer(dd[i] * fy[:, j] * fz[:, k], Xv)
                            J[:, :, 1] += np.outer(fx[:, i] * dd[j] * fz[:, k], Xv)
                            J[:, :, 2] += np.outer(fx[:, i] * fy[:, j] * dd[k], Xv)
                            xyz += np.outer(fx[:, i] * fy[:, j] * fz[:, k], Xv)
                det = np.linalg.det(J); Jinv = np.linalg.inv(J)
                m = (1.0 if sd else -1.0) * Jinv[:, d, :]
                ln = np.linalg.norm(m, axis=1)
                n = m / ln[:, None]
                JxW = np.abs(det) * ln * Wf
                h = np.sqrt(JxW.sum())
                gph = np.einsum("qed,qea->qda", Jinv, dNu)       # physical gradients of the velocity shape functions
                dn = np.einsum("qda,qd->qa", gph, n)             # their normal derivatives
                gq = np.array([gfun(x) for x in xyz])
                all_pts.append(xyz); all_g.append(gq)
                nun = (pu + 1) ** 3
                iu = np.array([(pu * cx + a) + ndu[0] * ((pu * cy + b) + ndu[1] * (pu * cz + c))
                               for c in range(pu + 1) for b in range(pu + 1) for a in range(pu + 1)])
                ip = np.array([(pp * cx + a) + ndp[0] * ((pp * cy + b) + ndp[1] * (pp * cz + c))
                               for c in range(pp + 1) for b in range(pp + 1) for a in range(pp + 1)])
                if dgp:
                    ip = (cx + ncell[0] * (cy + ncell[1] * cz)) * Np_.shape[1] + np.arange(Np_.shape[1])
                NN = np.einsum("q,qa,qb->ab", JxW, Nu_, Nu_)
                NdN = np.einsum("q,qa,qb->ab", JxW, Nu_, dn)     # v-value x normal derivative of u
                for c in range(3):
                    rows = c * NU + iu
                    Auu[np.ix_(rows, rows)] += -nu * NdN + (g1 / h) * NN - nu * NdN.T
                    for c2_ in range(3):
                        cols = c2_ * NU + iu
                        Auu[np.ix_(rows, cols)] += (g2 / h) * np.einsum("q,qa,qb->ab", JxW * n[:, c] * n[:, c2_], Nu_, Nu_)
                    Aup[np.ix_(rows, ip)] += np.einsum("q,qa,qb->ab", JxW * n[:, c], Nu_, Np_)
                    Apu[np.ix_(ip, rows)] += -np.einsum("q,qb,qa->ba", JxW * n[:, c], Np_, Nu_)
                    gn = np.einsum("qd,qd->q", gq, n)
                    Fu[rows] += np.einsum("q,qa->a", JxW * ((g1 / h) * gq[:, c] + (g2 / h) * n[:, c] * gn), Nu_) \
                        - nu * np.einsum("q,qa->a", JxW * gq[:, c], dn)
                Fp[ip] += -np.einsum("q,qb->b", JxW * np.einsum("qd,qd->q", gq, n), Np_)
                del nun
    con = np.zeros(ndu[::-1], dtype=bool)
    if mask & 1: con[:, :, 0] = True
    if mask & 2: con[:, :, -1] = True
    if mask & 4: con[:, 0, :] = True
    if mask & 8: con[:, -1, :] = True
    if mask & 16: con[0, :, :] = True
    if mask & 32: con[-1, :, :] = True
    con3 = np.tile(con.reshape(-1), 3)
    Auu[con3, :] = 0; Auu[:, con3] = 0; Aup[con3, :] = 0; Apu[:, con3] = 0; Fu[con3] = 0
    return Auu, Aup, Apu, Fu, Fp, np.concatenate(all_pts), np.concatenate(all_g)


def stokes_apply(K, M, B, nu, U, P):
    """U [3, NU], P [NP] -> (nu K u - B^T p, B u, M u)"""
    ou = np.stack([nu * K @ U[c] - B[c].T @ P for c in range(3)])
    op = sum(B[c] @ U[c] for c in range(3))
    mu = np.stack([M @ U[c] for c in range(3)])
    return ou, op, mu


def block_index(nt, nv, it, v, d):  # BlockSlice, variable-major (fe_time.h:956-967)
    return it * (nv * nt) + v * nt + d


def main():
    cases = [
        # name, ncell, lower, upper, jitter, mask, nu, time kind, r, tau
        ("stokes_cart_2x2x2", (2, 2, 2), (0, 0, 0), (1, 1.5, 0.7), 0.0, 63, 1.0, "cg", 2, 0.1),
        ("stokes_pert_2x3x2", (2, 3, 2), (0, 0, 0), (1, 1, 1), 0.15, 63, 0.05, "dg", 1, 1 / 16),
        ("stokes_free_3x2x2", (3, 2, 2), (-1, -1, -1), (1, 1, 1), 0.1, 0b010011, 2.5, "cg", 1, 0.25),
    ]
    for (name, ncell, lo, up, jit, mask, nu, kind, r, tau) in cases:
        rng = np.random.default_rng(sum(map(ord, name)))
        verts = structured_vertices(ncell, lo, up, jit, seed=23)
        K, M, B = dense_stokes(2, ncell, verts, mask)
        NU, NP = K.shape[0], B.shape[1]
        At, Bt = time_weights_single(kind, r, tau)  # single step: nt x nt
        nt = At.shape[0]
        # get_fe_time_weights_stokes layout (one time step at once): 2 nt x 2 nt
        nb = 2 * nt
        Alpha = np.zeros((nb, nb)); Beta = np.zeros((nb, nb))
        for iv in range(2):
            for jv in range(2):
                if not (iv == 1 and jv == 1):
                    for a in range(nt):
                        for b in range(nt):
                            Alpha[block_index(nt, 2, 0, iv, a), block_index(nt, 2, 0, jv, b)] = At[a, b]
        for a in range(nt):
            for b in range(nt):
                Beta[block_index(nt, 2, 0, 0, a), block_index(nt, 2, 0, 0, b)] = Bt[a, b]
        U = rng.uniform(-1, 1, size=(nt, 3, NU)); P = rng.uniform(-1, 1, size=(nt, NP))
        DU = np.zeros((nt, 3, NU)); DP = np.zeros((nt, NP))
        SU = np.zeros((nt, 3, NU)); SP = np.zeros((nt, NP)); MU = np.zeros((nt, 3, NU))
        for i in range(nt):
            ou, op, mu = stokes_apply(K, M, B, nu, U[i], P[i])
            SU[i], SP[i], MU[i] = ou, op, mu
            col = block_index(nt, 2, 0, 0, i)
            for j in range(nt):
                DU[j] += Alpha[block_index(nt, 2, 0, 0, j), col] * ou + Beta[block_index(nt, 2, 0, 0, j), col] * mu
                DP[j] += Alpha[block_index(nt, 2, 0, 1, j), col] * op
        np.savez_compressed(os.path.join(HERE, name + ".npz"), ncell=np.array(ncell), vertices=verts,
                            mask=mask, nu=nu, nt=nt, Alpha=Alpha, Beta=Beta, U=U, P=P, DU=DU, DP=DP,
                            SU=SU, SP=SP, MU=MU)
        print(name, "NU", NU, "NP", NP, "nt", nt, "|K|", np.abs(K).max(), "|B|", np.abs(B).max(),
              "B*1", np.abs(sum(B[c] @ np.ones(NU) for c in range(3))).max() if mask == 0 else "-")


def main_dgp():
    """FE_DGP(1) pressure (the reference's dGPressure = true, tests/tp_03stokes.cc:83-86): cell operator, Nitsche faces and the
    functional of the Dirichlet data"""
    gfun = lambda x: np.array([np.sin(1.3 * x[0] + 0.4 * x[1]) + x[2], np.cos(0.7 * x[1] - x[2]) * x[0], 0.5 + x[0] * x[1] - 0.3 * x[2] ** 2])  # noqa: E731
    cases = [
        ("stokes_dgp_cart_2x2x2", (2, 2, 2), (0, 0, 0), (1, 1.5, 0.7), 0.0, 63, 0, 1.0),
        ("stokes_dgp_pert_2x3x2", (2, 3, 2), (0, 0, 0), (1, 1, 1), 0.15, 0b001100, 0b110011, 0.05),
    ]
    for (name, ncell, lo, up, jit, mask, weak, nu) in cases:
        rng = np.random.default_rng(sum(map(ord, name)))
        verts = structured_vertices(ncell, lo, up, jit, seed=31)
        K, M, B = dense_stokes(2, ncell, verts, mask, dgp=True)
        NU, NP = K.shape[0], B.shape[1]
        U = rng.uniform(-1, 1, size=(3, NU)); P = rng.uniform(-1, 1, size=NP)
        ou, op, mu = stokes_apply(K, M, B, nu, U, P)
        out = dict(ncell=np.array(ncell), vertices=verts, mask=mask, weak=weak, nu=nu, penalty1=20.0, penalty2=10.0, U=U, P=P, MU=mu)
        if weak:
            Auu, Aup, Apu, Fu, Fp, pts, gq = dense_nitsche(2, ncell, verts, mask, weak, nu, 20.0, 10.0, gfun, dgp=True)
            ou = ou + (Auu @ U.reshape(-1) + Aup @ P).reshape(3, NU)
            op = op + Apu @ U.reshape(-1)
            out.update(FU=Fu.reshape(3, NU), FP=Fp, face_points=pts, G=gq)
        out.update(SU=ou, SP=op)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "NU", NU, "NP", NP, "|B|", np.abs(B).max())


def main_nitsche():
    """weak-boundary fixtures: StokesMatrixFreeOperator::vmult with Nitsche faces (LoopType::Full) and
    StokesNitscheMatrixFreeOperator::vmult for a smooth Dirichlet function"""
    gfun = lambda x: np.array([np.sin(1.3 * x[0] + 0.4 * x[1]) + x[2], np.cos(0.7 * x[1] - x[2]) * x[0], 0.5 + x[0] * x[1] - 0.3 * x[2] ** 2])  # noqa: E731
    cases = [
        # name, ncell, lower, upper, jitter, strong mask, weak mask, nu, penalty1, penalty2
        ("stokes_nitsche_cart_2x2x2", (2, 2, 2), (0, 0, 0), (1, 1.5, 0.7), 0.0, 0b001100, 0b110011, 1.0, 20.0, 10.0),
        ("stokes_nitsche_pert_2x3x2", (2, 3, 2), (0, 0, 0), (1, 1, 1), 0.15, 0, 0b111111, 0.05, 20.0, 10.0),
        ("stokes_nitsche_pert_3x2x2", (3, 2, 2), (-1, -1, -1), (1, 1, 1), 0.1, 0b010000, 0b100110, 2.5, 15.0, 4.0),
    ]
    for (name, ncell, lo, up, jit, mask, weak, nu, pen1, pen2) in cases:
        rng = np.random.default_rng(sum(map(ord, name)))
        verts = structured_vertices(ncell, lo, up, jit, seed=29)
        K, M, B = dense_stokes(2, ncell, verts, mask)
        Auu, Aup, Apu, Fu, Fp, pts, gq = dense_nitsche(2, ncell, verts, mask, weak, nu, pen1, pen2, gfun)
        NU, NP = K.shape[0], B.shape[1]
        U = rng.uniform(-1, 1, size=(3, NU)); P = rng.uniform(-1, 1, size=NP)
        ou, op, _ = stokes_apply(K, M, B, nu, U, P)
        SU = ou + (Auu @ U.reshape(-1) + Aup @ P).reshape(3, NU)
        SP = op + Apu @ U.reshape(-1)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), ncell=np.array(ncell), vertices=verts, mask=mask, weak=weak, nu=nu,
                            penalty1=pen1, penalty2=pen2, U=U, P=P, SU=SU, SP=SP, FU=Fu.reshape(3, NU), FP=Fp, face_points=pts, G=gq)
        print(name, "NU", NU, "NP", NP, "face points", len(pts), "|Auu|", np.abs(Auu).max(), "sym", np.abs(Auu - Auu.T).max(),
              "Aup + Apu^T", np.abs(Aup + Apu.T).max())


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "nitsche":
        main_nitsche()
    elif len(sys.argv) > 1 and sys.argv[1] == "dgp":
        main_dgp()
    else:
        main()
        main_nitsche()
        main_dgp()
