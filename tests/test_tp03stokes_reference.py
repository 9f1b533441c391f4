"""Pins the Stokes oracle's ingredients to ABSOLUTE numbers the reference itself holds.

tests/tp_03stokes.output (committed as the data golden tests/golden/tp_03stokes.output) is the reference's space-time convergence
study of the instationary Stokes problem in 2D (tests/tp_03stokes.cc with tests/json/tf01stokes.json / tf02stokes.json and
tests/json/stokes.json): velocity FE_Q(k+1)^2, pressure FE_DGP(k) (dGPressure = true), dG(k) / cG(k) in time, one time step per solve,
viscosity 1, homogeneous Dirichlet on the whole boundary, pressure with zero mean; errors of u in L-inf(L-inf), L2(L2), L2(H1-semi),
L2(Hdiv-semi) and of p in L-inf(L-inf), L2(L2), L2(H1-semi), to six digits.

This test re-derives rows of those tables with the ORACLE's ingredients (1D shape tables, Gauss rules, temporal matrices) and the
reference's recipe, solving every slab system directly (the reference's GMRES stops at 1e-12):
  tests/tp_03stokes.cc:79-106       FE_Q(k+1)^dim x FE_DGP(k), QGauss(k+2); tau = 2^-(refinement+1) on the unit square
  include/operators.h:1525-1575     pressure.submit_value(div u); velocity.submit_gradient(nu grad u - p I)
  include/operators.h:825-867       SystemMatrixStokes::vmult: K_S scattered with Alpha (both variables), M with Beta (velocity)
  include/fe_time.h:1242-1285       get_fe_time_weights_stokes (no pressure-pressure block; Gamma on the cG pressure rows)
  tests/tp_03stokes.cc:238-246      rhs_uK / rhs_uM (cG: Gamma, Zeta; dG: 0, Gamma)
  include/time_integrators.h:73-111 assemble_force per variable (the pressure load is zero)
  include/exact_solution.h:199-325  exact velocity / pressure and the force
  tests/tp_03stokes.cc:1047-1062    the pressure of every time dof is shifted to zero mean after the solve
  include/exact_solution.h:503-649  ErrorCalculator: QGauss(k+1) in time; QGauss(k+2) per direction for u, QGauss(k+1) for p;
                                    vector norms as VectorTools::integrate_difference defines them
  tests/tp_03stokes.cc:1079-1083    the reference's L-inf(p) column is max(pressure error of the LAST step, velocity L-inf so far)
What this pins: the Q2 / Q3 shape tables, the Gauss rules, the sign and block structure of [nu K, -B^T; B, 0], the vector mass,
the Stokes temporal matrices and the right-hand side recipe.  The pressure space of the golden (FE_DGP) is not the one of the 3D
oracle / kernels (FE_Q(k), BASELINE configs[4]); the second test closes that gap: on a Cartesian mesh the oracle's 3D operator is the
Kronecker form built from the same 1D tables, with the mixed Q1 x Q2 matrices in place of the DGP ones."""
import os
import re

import numpy as np
import pytest
import scipy.linalg

from test_tp01_reference import lagrange_eval, matrices_1d

PI = np.pi


def exact_u(x, y, t):
    s = np.sin(t)
    return (np.cos(PI * y) * s * np.sin(PI * x) ** 2 * np.sin(PI * y), -np.cos(PI * x) * s * np.sin(PI * x) * np.sin(PI * y) ** 2)


def exact_grad_u(x, y, t):
    ps = PI * np.sin(t)
    sx, sy, cx, cy = np.sin(PI * x), np.sin(PI * y), np.cos(PI * x), np.cos(PI * y)
    return ((2 * ps * cx * sx * cy * sy, ps * (sx * sx * cy * cy - sx * sx * sy * sy)),
            (ps * (sx * sx - cx * cx) * sy * sy, -2 * ps * cx * sx * cy * sy))


def exact_p(x, y, t):
    return np.cos(PI * x) * np.cos(PI * y) * np.sin(t) * np.sin(PI * x) * np.sin(PI * y)


def exact_grad_p(x, y, t):
    ps = PI * np.sin(t)
    sx, sy, cx, cy = np.sin(PI * x), np.sin(PI * y), np.cos(PI * x), np.cos(PI * y)
    return (ps * (cx * cx - sx * sx) * cy * sy, ps * (cy * cy - sy * sy) * cx * sx)


def force(x, y, t, nu=1.0):
    st, ct = np.sin(t), np.cos(t)
    sx, sy, cx, cy = np.sin(PI * x), np.sin(PI * y), np.cos(PI * x), np.cos(PI * y)
    f0 = sy * (PI * (1.0 - 2.0 * PI * nu) * cx * cx * cy * st + cy * (ct + PI * (-1.0 + 6.0 * PI * nu) * st) * sx * sx)
    f1 = sx * (cx * (PI * (-2.0 * PI * nu + (1.0 + 4.0 * PI * nu) * np.cos(2.0 * PI * y)) * st - ct * sy * sy))
    return f0, f1


def stokes_time_weights(o, ttype, k, tau):
    """get_fe_time_weights_stokes for one time step: (Alpha, Beta, Gamma, Zeta) in the (variable, time dof) block order"""
    A, B, G, Z = o.time_weights(ttype, k, tau, 1)
    nt = A.shape[0]
    idx = lambda v, a: v * nt + a  # noqa: E731  BlockSlice::index, variable-major, one step
    Al, Be = np.zeros((2 * nt, 2 * nt)), np.zeros((2 * nt, 2 * nt))
    Ga, Ze = np.zeros((2 * nt, G.shape[1])), np.zeros((2 * nt, Z.shape[1]))
    for a in range(nt):
        for b in range(nt):
            for iv in range(2):
                for jv in range(2):
                    if not (iv == 1 and jv == 1):
                        Al[idx(iv, a), idx(jv, b)] = A[a, b]
            Be[idx(0, a), idx(0, b)] = B[a, b]
        Ga[idx(0, a)] = G[a]
        if ttype == o.CGP:
            Ga[idx(1, a)] = G[a]
        Ze[idx(0, a)] = Z[a]
    return Al, Be, Ga, Ze


def dgp_monomials(k):
    """exponents (i, j) of the basis (xi - 1/2)^i (eta - 1/2)^j, i + j <= k: a basis of FE_DGP(k) on the reference cell
    (any basis gives the same discrete solution; the first function is the constant)"""
    return [(i, d - i) for d in range(k + 1) for i in range(d, -1, -1)]


def convergence_row(o, ttype, k, refinement, nu=1.0):
    pu = k + 1
    n = 2 ** refinement
    h = 1.0 / n
    nd = pu * n + 1
    tau = 2.0 ** -(refinement + 1)
    M1, K1 = matrices_1d(o, pu, n)  # QGauss(pu + 1) = QGauss(k + 2), as quads_u
    M2 = np.kron(M1, M1)
    K2 = np.kron(K1, M1) + np.kron(M1, K1)  # index iy * nd + ix
    idx = np.arange(nd * nd).reshape(nd, nd)
    free = idx[1:-1, 1:-1].ravel()
    nf = len(free)
    Kf, Mf = K2[np.ix_(free, free)], M2[np.ix_(free, free)]
    # divergence coupling with the DGP(k) pressure: B[c][cell * nq + j, node] = int_cell q_j d phi_node / d x_c
    S, D = o.shape_tables(pu)  # [q, a] at the QGauss(pu + 1) points of the reference cell
    xq, wq = o.gauss(pu + 1)
    mono = dgp_monomials(k)
    npq = len(mono)
    NP = n * n * npq
    Bx, By = np.zeros((NP, nd * nd)), np.zeros((NP, nd * nd))
    for j, (ei, ej) in enumerate(mono):
        # 1D factors: int (xi - 1/2)^e phi_a' d xi (the 1 / h of the derivative cancels the h of the measure), h int (.)^e phi_a d xi
        dx = (D.T * wq) @ (xq - 0.5) ** ei
        mx = h * (S.T * wq) @ (xq - 0.5) ** ei
        dy = (D.T * wq) @ (xq - 0.5) ** ej
        my = h * (S.T * wq) @ (xq - 0.5) ** ej
        for cy in range(n):
            for cx in range(n):
                row = (cy * n + cx) * npq + j
                nodes = idx[pu * cy:pu * cy + pu + 1, pu * cx:pu * cx + pu + 1]
                Bx[row, nodes.ravel()] += np.outer(my, dx).ravel()
                By[row, nodes.ravel()] += np.outer(dy, mx).ravel()
    Bf = np.hstack([Bx[:, free], By[:, free]])  # acts on (u_x free, u_y free)
    NU = 2 * nf
    Z2 = np.zeros((nf, nf))
    KS_uu = nu * np.block([[Kf, Z2], [Z2, Kf]])
    MM = np.block([[Mf, Z2], [Z2, Mf]])
    Al, Be, Ga, Ze = stokes_time_weights(o, ttype, k, tau)
    A1 = o.time_weights(ttype, k, tau, 1)[0]
    G1 = o.time_weights(ttype, k, tau, 1)[2]
    nt = A1.shape[0]
    # SystemMatrixStokes (operators.h:825-867): rows / columns ordered (u time dofs, p time dofs)
    N = nt * (NU + NP)
    sysm = np.zeros((N, N))
    ub = lambda a: slice(a * NU, (a + 1) * NU)  # noqa: E731
    pb = lambda a: slice(nt * NU + a * NP, nt * NU + (a + 1) * NP)  # noqa: E731
    for a in range(nt):
        for b in range(nt):
            sysm[ub(a), ub(b)] += Al[a, b] * KS_uu + Be[a, b] * MM
            sysm[ub(a), pb(b)] += Al[a, b] * (-Bf.T)        # velocity rows: the source column of K_S is the pair (u_b, p_b)
            sysm[pb(a), ub(b)] += Al[nt + a, b] * Bf         # pressure rows: Alpha(idx(1, a), idx(0, b)) (B u_b)
    # the pressure is determined up to a constant: pin the constant mode of cell 0 in every time dof, shift to zero mean afterwards
    keep = np.ones(N, dtype=bool)
    for a in range(nt):
        keep[nt * NU + a * NP] = False
    lu = scipy.linalg.lu_factor(sysm[np.ix_(keep, keep)])
    rK, rM = (Ga, Ze) if ttype == o.CGP else (np.zeros_like(Ga), Ga)  # tests/tp_03stokes.cc:243-244

    def load_vector(t):
        """VectorTools::create_right_hand_side with stokes::RHSFunction, QGauss(k + 2)"""
        F = np.zeros((2, nd * nd))
        W = h * h * np.outer(wq, wq)
        for cy in range(n):
            for cx in range(n):
                X, Y = h * (cx + xq)[None, :], h * (cy + xq)[:, None]
                f0, f1 = force(X, Y, t, nu)
                nodes = idx[pu * cy:pu * cy + pu + 1, pu * cx:pu * cx + pu + 1].ravel()
                for c, f in enumerate((f0, f1)):
                    F[c, nodes] += np.einsum("yx,ya,xb->ab", W * f, S, S).ravel()
        return np.concatenate([F[0, free], F[1, free]])

    tq_int = o.gauss_radau_right(k + 1) if ttype == o.DG else o.gauss_lobatto(k + 1)
    et, ewt = o.gauss(k + 1)
    Ltime, _ = lagrange_eval(tq_int, et)
    gll = o.gauss_lobatto(pu + 1)
    eu, ewu = o.gauss(k + 2)  # ErrorCalculator(type, fe_degree, fe_u.tensor_degree()): QGauss(space_degree + 1)
    ep, ewp = o.gauss(k + 1)
    Eu, dEu = lagrange_eval(gll, eu)

    def errors_u(uf, t):
        U = np.zeros((2, nd * nd))
        U[0, free], U[1, free] = uf[:nf], uf[nf:]
        U = U.reshape(2, nd, nd)
        l2 = h1 = hdiv = l8 = 0.0
        W = h * h * np.outer(ewu, ewu)
        for cy in range(n):
            for cx in range(n):
                X, Y = h * (cx + eu)[None, :], h * (cy + eu)[:, None]
                ue, ge = exact_u(X, Y, t), exact_grad_u(X, Y, t)
                div = 0.0
                for c in range(2):
                    loc = U[c, pu * cy:pu * cy + pu + 1, pu * cx:pu * cx + pu + 1]
                    uh, ux, uy = Eu @ loc @ Eu.T, (Eu @ loc @ dEu.T) / h, (dEu @ loc @ Eu.T) / h
                    l2 += np.sum(W * (uh - ue[c]) ** 2)
                    h1 += np.sum(W * ((ux - ge[c][0]) ** 2 + (uy - ge[c][1]) ** 2))
                    l8 = max(l8, np.abs(uh - ue[c]).max())
                    div = div + ((ux - ge[c][0]) if c == 0 else (uy - ge[c][1]))
                hdiv += np.sum(W * div ** 2)
        return l2, l8, h1, hdiv

    def errors_p(pf, t):
        P = pf.reshape(n, n, npq)
        l2 = h1 = l8 = 0.0
        W = h * h * np.outer(ewp, ewp)
        XI, ETA = (ep - 0.5)[None, :], (ep - 0.5)[:, None]
        for cy in range(n):
            for cx in range(n):
                ph = np.zeros((len(ep), len(ep))); px = np.zeros_like(ph); py = np.zeros_like(ph)
                for j, (ei, ej) in enumerate(mono):
                    ph += P[cy, cx, j] * XI ** ei * ETA ** ej
                    if ei:
                        px += P[cy, cx, j] * ei * XI ** (ei - 1) * ETA ** ej / h
                    if ej:
                        py += P[cy, cx, j] * ej * XI ** ei * ETA ** (ej - 1) / h
                X, Y = h * (cx + ep)[None, :], h * (cy + ep)[:, None]
                pe, ge = exact_p(X, Y, t), exact_grad_p(X, Y, t)
                l2 += np.sum(W * (ph - pe) ** 2)
                h1 += np.sum(W * ((px - ge[0]) ** 2 + (py - ge[1]) ** 2))
                l8 = max(l8, np.abs(ph - pe).max())
        return l2, l8, h1

    # mean of a DGP function: QGauss(k + 1) integrates the monomials of degree <= k exactly
    mq, mw = o.gauss(k + 1)
    mono_mean = np.array([np.sum(np.outer(mw, mw) * ((mq - 0.5)[None, :] ** ei) * ((mq - 0.5)[:, None] ** ej)) for ei, ej in mono])
    prev_u = np.zeros(NU)
    prev_p = np.zeros(NP)
    time = 0.0
    acc = dict(l2=0.0, l8=-1.0, h1=0.0, hdiv=0.0, l2p=0.0, l8p=-1.0, h1p=0.0)
    while time < 1.0 - 1e-12:
        rhs = np.zeros(N)
        KSprev_u = KS_uu @ prev_u - Bf.T @ prev_p
        KSprev_p = Bf @ prev_u
        for a in range(nt):
            rhs[ub(a)] = rK[a, 0] * KSprev_u + rM[a, 0] * (MM @ prev_u)
            rhs[pb(a)] = rK[nt + a, 0] * KSprev_p
        for j, xi in enumerate(tq_int):  # assemble_force, velocity variable (the pressure load is zero)
            F = load_vector(time + tau * xi)
            if ttype == o.DG:
                rhs[ub(j)] += A1[j, j] * F
            elif j == 0:
                for i in range(nt):
                    rhs[ub(i)] += -G1[i, 0] * F
            else:
                rhs[ub(j - 1)] += A1[j - 1, j - 1] * F
        sol = np.zeros(N)
        sol[keep] = scipy.linalg.lu_solve(lu, rhs[keep])
        xu = [sol[ub(a)] for a in range(nt)]
        xp = []
        for a in range(nt):  # zero mean (tests/tp_03stokes.cc:1047-1062)
            p = sol[pb(a)].reshape(n * n, npq).copy()
            mean = h * h * np.sum(p @ mono_mean)
            p[:, 0] -= mean
            xp.append(p.ravel())
        last_l8p = -1.0
        for q in range(k + 1):  # ErrorCalculator::evaluate_error
            if ttype == o.DG:
                uf = sum(Ltime[q, i] * xu[i] for i in range(nt))
                pf = sum(Ltime[q, i] * xp[i] for i in range(nt))
            else:
                uf = Ltime[q, 0] * prev_u + sum(Ltime[q, i] * xu[i - 1] for i in range(1, k + 1))
                pf = Ltime[q, 0] * prev_p + sum(Ltime[q, i] * xp[i - 1] for i in range(1, k + 1))
            t = time + tau * et[q]
            l2, l8, h1, hdiv = errors_u(uf, t)
            acc["l2"] += tau * ewt[q] * l2
            acc["h1"] += tau * ewt[q] * h1
            acc["hdiv"] += tau * ewt[q] * hdiv
            acc["l8"] = max(acc["l8"], l8)
            l2p, l8p, h1p = errors_p(pf, t)
            acc["l2p"] += tau * ewt[q] * l2p
            acc["h1p"] += tau * ewt[q] * h1p
            last_l8p = max(last_l8p, l8p)
        acc["l8p"] = max(last_l8p, acc["l8"])  # tests/tp_03stokes.cc:1079-1083 (sic)
        prev_u, prev_p = xu[-1], xp[-1]
        time += tau
    return (acc["l8"], np.sqrt(acc["l2"]), np.sqrt(acc["h1"]), np.sqrt(acc["hdiv"]), acc["l8p"], np.sqrt(acc["l2p"]), np.sqrt(acc["h1p"]))


def golden_tables(golden_dir):
    """-> list of (k, rows) in file order (four dG tables, then four cG tables); rows = [(cells, t_dofs, 7 error values)]"""
    tables = []
    with open(os.path.join(golden_dir, "tp_03stokes.output"), encoding="utf-8") as f:
        lines = f.read().splitlines()
    for n, line in enumerate(lines):
        m = re.match(r"^Convergence table k=(\d+)$", line)
        if not m:
            continue
        rows = []
        for r in lines[n + 2:n + 6]:
            t = r.split()
            if len(t) < 6 or not t[0].isdigit():
                break
            vals = [v for v in t[5:] if re.match(r"^\d\.\d+e[-+]\d+$", v)]
            rows.append((int(t[0]), int(t[2]), [float(v) for v in vals[:7]]))
        tables.append((int(m.group(1)), rows))
    return tables


# first run of the golden: tests/json/tf01stokes.json (dG(k), k = 1..4), second: tf02stokes.json (cG(k), k = 1..4)
CASES = [("DG", 1, 0, (1, 2, 3)), ("DG", 2, 1, (1, 2)), ("CG", 1, 4, (1, 2, 3)), ("CG", 2, 5, (1, 2))]


@pytest.mark.parametrize("kind,k,table,refinements", CASES)
def test_stokes_convergence_rows_of_tp03(oracle_mod, golden_dir, kind, k, table, refinements):
    o = oracle_mod
    gk, rows = golden_tables(golden_dir)[table]
    ttype = o.DG if kind == "DG" else o.CGP
    assert gk == k and rows[0][1] == (k + 1 if kind == "DG" else k), (gk, rows[0])
    names = ("Linf(u)", "L2L2(u)", "L2H1(u)", "L2Hdiv(u)", "Linf(p)", "L2L2(p)", "L2H1(p)")
    for ref in refinements:
        cells, _, gold = rows[ref - 1]
        assert cells == 4 ** ref
        got = convergence_row(o, ttype, k, ref)
        for name, g, w in zip(names, got, gold):
            digits = 4 if name == "L2Hdiv(u)" else 5  # the Hdiv column is printed with one digit less
            ulp = 10.0 ** (np.floor(np.log10(w)) - digits)
            # to the printed digits; the pointwise maxima to one (velocity) / two (pressure, the multiplier) units of the last
            # digit: the reference's GMRES stops at a relative residual of 1e-12, every slab system is solved directly here
            slack = 2.01 if name == "Linf(p)" else (1.01 if name == "Linf(u)" else 0.51)
            assert abs(g - w) <= slack * ulp + 1e-8 * w, (kind, k, ref, name, g, w)


@pytest.mark.parametrize("nc,upper,nu", [((3, 2, 2), (1.0, 0.75, 1.5), 0.3), ((2, 2, 1), (2.0, 1.0, 0.5), 1.7)])
def test_oracle_3d_stokes_operator_is_the_kronecker_form_of_the_same_1d_tables(oracle_mod, nc, upper, nu):
    """closes the chain to the operator the HIP kernels are compared with (oracle/stfem_oracle_stokes.c, FE_Q(2)^3 x FE_Q(1)): on a
    Cartesian mesh  nu K (x) I_3,  the vector mass  and  B_c = (q, d u_c / d x_c)  are Kronecker products of 1D matrices built from
    the shape tables and Gauss rule that reproduce the reference's tables above - with the mixed Q1 x Q2 matrices
    N = int psi_j phi_a, C = int psi_j phi_a' in place of the DGP ones."""
    import importlib
    o = oracle_mod
    stfem = importlib.import_module("dealii-stfem_amd")
    verts = stfem.mesh_vertices(nc, (0, 0, 0), upper)
    orc = o.StokesOracle(nc, verts, 0, nu)
    Su, Du = o.shape_tables(2)        # [q, a] at QGauss(3)
    Sp, _ = o.shape_tables(1, 3)
    _, w = o.gauss(3)
    m1, k1, n1, c1 = [], [], [], []
    for d in range(3):
        n, L = nc[d], upper[d]
        M1, K1 = matrices_1d(o, 2, n)
        m1.append(M1 * L)
        k1.append(K1 / L)
        h = L / n
        N = np.zeros((n + 1, 2 * n + 1)); Cm = np.zeros((n + 1, 2 * n + 1))
        for c in range(n):
            N[c:c + 2, 2 * c:2 * c + 3] += h * (Sp.T * w) @ Su
            Cm[c:c + 2, 2 * c:2 * c + 3] += (Sp.T * w) @ Du
        n1.append(N); c1.append(Cm)
    kron3 = lambda z, y, x: np.kron(z, np.kron(y, x))  # noqa: E731  index ix + nx (iy + ny iz)
    K3 = kron3(m1[2], m1[1], k1[0]) + kron3(m1[2], k1[1], m1[0]) + kron3(k1[2], m1[1], m1[0])
    M3 = kron3(m1[2], m1[1], m1[0])
    B = [kron3(n1[2], n1[1], c1[0]), kron3(n1[2], c1[1], n1[0]), kron3(c1[2], n1[1], n1[0])]
    rng = np.random.default_rng(4)
    U, P = rng.uniform(-1, 1, (3, orc.n_u)), rng.uniform(-1, 1, orc.n_p)
    ou, op = orc.apply(U, P)
    want_u = np.stack([nu * K3 @ U[c] - B[c].T @ P for c in range(3)])
    want_p = sum(B[c] @ U[c] for c in range(3))
    assert np.abs(ou - want_u).max() <= 1e-12 * np.abs(want_u).max()
    assert np.abs(op - want_p).max() <= 1e-12 * np.abs(want_p).max()
    mu, _ = orc.apply(U, P, 0.0, 1.0)
    assert np.abs(mu - np.stack([M3 @ U[c] for c in range(3)])).max() <= 1e-13 * np.abs(mu).max()
