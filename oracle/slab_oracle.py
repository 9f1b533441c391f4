"""CPU restatement (numpy, dense direct solves) of the reference's heat convergence test in 3D
(tests/tp_01.cc with space_time_conv_test, ProblemType::heat; include/time_integrators.h:73-111, 300-321;
include/exact_solution.h:27-81, 503-649).  TEST INFRASTRUCTURE ONLY.

The same recipe as tests/test_tp01_reference.py::convergence_row - which reproduces the reference's own 2D
numbers (tests/tp_01.output) to the printed digits - with one more space dimension: that test pins the recipe,
this module applies it in the dimension the HIP path works in.  Every slab system is solved directly, so the
result differs from an FGMRES solve by the solver tolerance (1e-12) only."""
import numpy as np

from . import oracle as o


def lagrange_eval(nodes, x):
    nodes = np.asarray(nodes, dtype=float)
    x = np.atleast_1d(np.asarray(x, dtype=float))
    n = len(nodes)
    L = np.ones((len(x), n))
    dL = np.zeros((len(x), n))
    for a in range(n):
        for m in range(n):
            if m != a:
                L[:, a] *= (x - nodes[m]) / (nodes[a] - nodes[m])
        for m in range(n):
            if m == a:
                continue
            t = np.full(len(x), 1.0 / (nodes[a] - nodes[m]))
            for l in range(n):
                if l != a and l != m:
                    t *= (x - nodes[l]) / (nodes[a] - nodes[l])
            dL[:, a] += t
    return L, dL


def matrices_1d(p, n):
    S, D = o.shape_tables(p)
    _, w = o.gauss(p + 1)
    h = 1.0 / n
    Mc = h * (S.T * w) @ S
    Kc = (1.0 / h) * (D.T * w) @ D
    nd = p * n + 1
    M = np.zeros((nd, nd))
    K = np.zeros((nd, nd))
    for c in range(n):
        sl = slice(p * c, p * c + p + 1)
        M[sl, sl] += Mc
        K[sl, sl] += Kc
    return M, K


def wave_convergence_row_3d(ttype, k, refinement, nsteps=2, frequency=1.0):
    """The same for the wave equation u_tt - laplace u = f (tests/tp_01.cc ProblemType::wave, include/time_integrators.h:343-459,
    include/exact_solution.h:147-197): slab system of fe_time.h:157-305 for u, velocity recovered block by block."""
    return heat_convergence_row_3d(ttype, k, refinement, nsteps, frequency, wave=True)


def heat_convergence_row_3d(ttype, k, refinement, nsteps=2, frequency=1.0, wave=False):
    """(L-inf L-inf, L2 L2, L2 H1-semi) of u = sin(2 pi f t) prod_d sin(2 pi f x_d) on the unit cube,
    FE_Q(k + 1) x {cG, dG}(k), 2^refinement cells per direction, tau = 2^-(refinement + 1)"""
    p = k + 1
    n = 2 ** refinement
    h = 1.0 / n
    nd = p * n + 1
    tau = 2.0 ** -(refinement + 1)
    M1, K1 = matrices_1d(p, n)
    f1 = slice(1, nd - 1)  # zero Dirichlet: interior nodes per direction
    Mi, Ki = M1[f1, f1], K1[f1, f1]
    M = np.kron(Mi, np.kron(Mi, Mi))
    K = np.kron(Ki, np.kron(Mi, Mi)) + np.kron(Mi, np.kron(Ki, Mi)) + np.kron(Mi, np.kron(Mi, Ki))
    nfree = (nd - 2) ** 3
    A1, B1, G1, Z1 = o.time_weights(ttype, k, tau, 1)
    ntd = k if ttype == o.CGP else k + 1
    nb = ntd * nsteps
    if wave:  # tests/tp_01.cc:143-158
        A, B, rK, rM, rV = o.time_weights_wave(ttype, k, tau, nsteps)
        Ainv = np.linalg.inv(A1)
        AixB, AixG, AixZ = Ainv @ B1, Ainv @ G1, Ainv @ Z1
        if ttype == o.DG:
            AixG = -AixG
        else:
            AixZ = -AixZ
    else:
        A, B, G, Z = o.time_weights(ttype, k, tau, nsteps)
        rK, rM = (G, Z) if ttype == o.CGP else (np.zeros_like(G), G)
    sysmat = np.kron(A, K) + np.kron(B, M)
    gll = o.gauss_lobatto(p + 1)
    xq, wq = o.gauss(p + 1)
    S, _ = o.shape_tables(p)
    w2 = 2 * np.pi * frequency

    def load_vector(t):
        amp = (2 * w2 ** 2 * np.sin(w2 * t)) if wave else (3 * (w2 ** 2) * np.sin(w2 * t) + w2 * np.cos(w2 * t))
        f = np.zeros(nd)
        for c in range(n):
            xs = h * (c + xq)
            f[p * c:p * c + p + 1] += h * (S.T * wq) @ np.sin(w2 * xs)
        fi = f[f1]
        return amp * np.einsum("i,j,k->ijk", fi, fi, fi).ravel()

    tq_int = o.gauss_radau_right(k + 1) if ttype == o.DG else o.gauss_lobatto(k + 1)
    et, ewt = o.gauss(k + 1)
    ex, ewx = o.gauss(k + 1)
    Ltime, _ = lagrange_eval(tq_int, et)
    E, dE = lagrange_eval(gll, ex)

    def spatial_errors(ufree, t):
        U = np.zeros((nd, nd, nd))  # [iz, iy, ix]
        U[1:-1, 1:-1, 1:-1] = ufree.reshape(nd - 2, nd - 2, nd - 2)
        l2 = h1 = l8 = 0.0
        st = np.sin(w2 * t)
        W = h ** 3 * np.einsum("i,j,k->ijk", ewx, ewx, ewx)
        for cz in range(n):
            for cy in range(n):
                for cx in range(n):
                    loc = U[p * cz:p * cz + p + 1, p * cy:p * cy + p + 1, p * cx:p * cx + p + 1]
                    uh = np.einsum("ac,bd,ef,cdf->abe", E, E, E, loc)        # [qz, qy, qx]
                    ux = np.einsum("ac,bd,ef,cdf->abe", E, E, dE, loc) / h
                    uy = np.einsum("ac,bd,ef,cdf->abe", E, dE, E, loc) / h
                    uz = np.einsum("ac,bd,ef,cdf->abe", dE, E, E, loc) / h
                    X = (h * (cx + ex))[None, None, :]
                    Y = (h * (cy + ex))[None, :, None]
                    Zc = (h * (cz + ex))[:, None, None]
                    sx, sy, sz = np.sin(w2 * X), np.sin(w2 * Y), np.sin(w2 * Zc)
                    cxs, cys, czs = np.cos(w2 * X), np.cos(w2 * Y), np.cos(w2 * Zc)
                    ue = st * sx * sy * sz
                    l2 += np.sum(W * (uh - ue) ** 2)
                    h1 += np.sum(W * ((ux - st * w2 * cxs * sy * sz) ** 2 + (uy - st * w2 * sx * cys * sz) ** 2 +
                                      (uz - st * w2 * sx * sy * czs) ** 2))
                    l8 = max(l8, np.abs(uh - ue).max())
        return l2, l8, h1

    prev = np.zeros(nfree)
    # v(0) = 2 pi f prod sin(2 pi f x_d) interpolated at the nodes
    xn = np.concatenate([h * (c + gll[:-1]) for c in range(n)] + [[1.0]])[f1]
    prev_v = w2 * np.einsum("i,j,k->ijk", np.sin(w2 * xn), np.sin(w2 * xn), np.sin(w2 * xn)).ravel()
    time, acc_l2, acc_l8, acc_h1 = 0.0, 0.0, -1.0, 0.0
    while time < 1.0 - 1e-12:
        rhs = np.zeros(nb * nfree)
        blk = lambda j: slice(j * nfree, (j + 1) * nfree)  # noqa: E731
        for j in range(nb):
            rhs[blk(j)] = rK[j, 0] * (K @ prev) + rM[j, 0] * (M @ prev)
            if wave:
                rhs[blk(j)] += rV[j, 0] * (M @ prev_v)
        for it in range(nsteps):
            for j, xi in enumerate(tq_int):
                F = load_vector(time + tau * it + tau * xi)
                if ttype == o.DG:
                    rhs[blk(it * ntd + j)] += A1[j, j] * F
                elif j == 0:
                    for i in range(ntd):
                        rhs[blk(it * ntd + i)] += -G1[i, 0] * F
                else:
                    rhs[blk(it * ntd + j - 1)] += A1[j - 1, j - 1] * F
        x = np.linalg.solve(sysmat, rhs).reshape(nb, nfree)
        for it in range(nsteps):
            prev_it = prev if it == 0 else x[ntd * it - 1]
            for q in range(k + 1):
                if ttype == o.DG:
                    uf = sum(Ltime[q, i] * x[it * ntd + i] for i in range(ntd))
                else:
                    uf = Ltime[q, 0] * prev_it + sum(Ltime[q, i] * x[it * ntd + i - 1] for i in range(1, k + 1))
                l2, l8, h1 = spatial_errors(uf, time + tau * it + tau * et[q])
                acc_l2 += tau * ewt[q] * l2
                acc_h1 += tau * ewt[q] * h1
                acc_l8 = max(acc_l8, l8)
        if wave:  # velocity recovery (time_integrators.h:429-446)
            v = np.zeros_like(x)
            for it in range(nsteps):
                sl = slice(it * ntd, (it + 1) * ntd)
                pu = prev if it == 0 else x[it * ntd - 1]
                v[sl] = AixB @ x[sl]
                if ttype == o.DG:
                    v[sl] += AixG @ pu[None, :]
                else:
                    pv = prev_v if it == 0 else v[it * ntd - 1]
                    v[sl] += AixG @ pv[None, :] + AixZ @ pu[None, :]
            prev_v = v[-1]
        prev = x[-1]
        time += nsteps * tau
    return acc_l8, np.sqrt(acc_l2), np.sqrt(acc_h1)


# ---------------------------------------------------------------------------------------------------------------------------------
# Instationary Stokes in 3D (BASELINE configs[4]: FE_Q(2)^3 x FE_Q(1), cG / dG in time).  The recipe of tests/tp_03stokes.cc, which
# tests/test_tp03stokes_reference.py pins to the reference's own 2D tables, in the dimension the HIP path works in:
#   operators.h:825-867 SystemMatrixStokes, fe_time.h:1242-1285 get_fe_time_weights_stokes, tp_03stokes.cc:238-246 right-hand-side
#   matrices, time_integrators.h:73-111 assemble_force (velocity only), tp_03stokes.cc:1047-1062 pressure shifted to zero mean,
#   exact_solution.h:503-649 ErrorCalculator (QGauss(k + 1) in time, QGauss(3) per direction for u, QGauss(2) for p).
# The reference's exact solution (exact_solution.h:199-325) is two-dimensional; this one is its 3D analogue: the velocity is the curl of
# psi e_z, psi = sin t (sin pi x sin pi y sin pi z)^2 (divergence-free, zero on the whole boundary), the pressure sin t cos pi x cos pi y
# cos pi z (zero mean), the force f = u_t - nu laplace u + grad p.
PI = np.pi


def stokes3d_exact_u(X, Y, Z, t):
    A = lambda s: np.sin(PI * s) ** 2               # noqa: E731
    B = lambda s: np.sin(PI * s) * np.cos(PI * s)   # noqa: E731
    st = np.sin(t)
    return (2 * PI * st * A(X) * B(Y) * A(Z), -2 * PI * st * B(X) * A(Y) * A(Z), np.zeros_like(X + Y + Z))


def stokes3d_exact_grad_u(X, Y, Z, t):
    A = lambda s: np.sin(PI * s) ** 2                     # noqa: E731
    dA = lambda s: PI * np.sin(2 * PI * s)                # noqa: E731
    B = lambda s: 0.5 * np.sin(2 * PI * s)                # noqa: E731
    dB = lambda s: PI * np.cos(2 * PI * s)                # noqa: E731
    c = 2 * PI * np.sin(t)
    z = np.zeros_like(X + Y + Z)
    return ((c * dA(X) * B(Y) * A(Z), c * A(X) * dB(Y) * A(Z), c * A(X) * B(Y) * dA(Z)),
            (-c * dB(X) * A(Y) * A(Z), -c * B(X) * dA(Y) * A(Z), -c * B(X) * A(Y) * dA(Z)),
            (z, z, z))


def stokes3d_exact_p(X, Y, Z, t):
    return np.sin(t) * np.cos(PI * X) * np.cos(PI * Y) * np.cos(PI * Z)


def stokes3d_force(X, Y, Z, t, nu):
    A = lambda s: np.sin(PI * s) ** 2                     # noqa: E731
    d2A = lambda s: 2 * PI * PI * np.cos(2 * PI * s)      # noqa: E731
    B = lambda s: 0.5 * np.sin(2 * PI * s)                # noqa: E731
    d2B = lambda s: -4 * PI * PI * B(s)                   # noqa: E731
    st, ct = np.sin(t), np.cos(t)
    lap1 = d2A(X) * B(Y) * A(Z) + A(X) * d2B(Y) * A(Z) + A(X) * B(Y) * d2A(Z)
    lap2 = d2B(X) * A(Y) * A(Z) + B(X) * d2A(Y) * A(Z) + B(X) * A(Y) * d2A(Z)
    sx, sy, sz, cx, cy, cz = np.sin(PI * X), np.sin(PI * Y), np.sin(PI * Z), np.cos(PI * X), np.cos(PI * Y), np.cos(PI * Z)
    f1 = 2 * PI * (ct * A(X) * B(Y) * A(Z) - nu * st * lap1) - PI * st * sx * cy * cz
    f2 = -2 * PI * (ct * B(X) * A(Y) * A(Z) - nu * st * lap2) - PI * st * cx * sy * cz
    f3 = -PI * st * cx * cy * sz + 0.0 * (X + Y + Z)
    return f1, f2, f3


def stokes_convergence_row_3d(ttype, k, refinement, nu=1.0, dg_pressure=False):
    """(u: L-inf L-inf, L2 L2, L2 H1-semi; p: L2 L2) of the solution above on the unit cube, FE_Q(2)^3 x FE_Q(1) x {cG, dG}(k), 2^refinement
    cells per direction, tau = 2^-(refinement + 1), homogeneous Dirichlet velocity on the whole boundary, one time step per solve.
    dg_pressure: FE_DGP(1) instead of FE_Q(1) (tests/tp_03stokes.cc:83-86, the reference's default)."""
    n = 2 ** refinement
    h = 1.0 / n
    tau = 2.0 ** -(refinement + 1)
    nc = (n, n, n)
    verts = np.array([[i * h, j * h, kk * h] for kk in range(n + 1) for j in range(n + 1) for i in range(n + 1)], dtype=float)
    so = o.StokesOracle(nc, verts, 0, nu, dg_pressure=dg_pressure)
    Nu, Np = so.n_u, so.n_p
    ndu, ndp = 2 * n + 1, n + 1
    ntot = 3 * Nu + Np
    K = np.zeros((ntot, ntot))
    M = np.zeros((3 * Nu, 3 * Nu))
    e = np.zeros(ntot)
    for j in range(ntot):
        e[j] = 1.0
        ou, op = so.apply(e[:3 * Nu], e[3 * Nu:], 1.0, 0.0)
        K[:3 * Nu, j], K[3 * Nu:, j] = ou.reshape(-1), op
        if j < 3 * Nu:
            M[:, j] = so.apply(e[:3 * Nu], np.zeros(Np), 0.0, 1.0)[0].reshape(-1)
        e[j] = 0.0
    iu = np.arange(ndu ** 3).reshape(ndu, ndu, ndu)
    free1 = iu[1:-1, 1:-1, 1:-1].ravel()
    free = np.concatenate([c * Nu + free1 for c in range(3)])
    nf = len(free)
    pidx = 3 * Nu + np.arange(Np)
    KS_uu, Bt, Bm = K[np.ix_(free, free)], K[np.ix_(free, pidx)], K[np.ix_(pidx, free)]   # nu K, -B^T, B
    MM = M[np.ix_(free, free)]
    A1, B1, G1, Z1 = o.time_weights(ttype, k, tau, 1)
    nt = A1.shape[0]
    NU, NP = nf, Np
    N = nt * (NU + NP)
    ub = lambda a: slice(a * NU, (a + 1) * NU)                     # noqa: E731
    pb = lambda a: slice(nt * NU + a * NP, nt * NU + (a + 1) * NP)  # noqa: E731
    sysm = np.zeros((N, N))
    for a in range(nt):
        for b in range(nt):
            sysm[ub(a), ub(b)] += A1[a, b] * KS_uu + B1[a, b] * MM
            sysm[ub(a), pb(b)] += A1[a, b] * Bt
            sysm[pb(a), ub(b)] += A1[a, b] * Bm
    keep = np.ones(N, dtype=bool)
    for a in range(nt):
        keep[nt * NU + a * NP] = False       # the pressure is determined up to a constant: pin one value, shift to zero mean afterwards
    import scipy.linalg
    lu = scipy.linalg.lu_factor(sysm[np.ix_(keep, keep)])
    # right-hand-side matrices (tests/tp_03stokes.cc:243-244): cG: Gamma on K_S (both variables), Zeta on M; dG: Gamma on M
    if ttype == o.CGP:
        rKu, rKp, rM = G1[:, 0], G1[:, 0], Z1[:, 0]
    else:
        rKu, rKp, rM = np.zeros(nt), np.zeros(nt), G1[:, 0]
    Su, _ = o.shape_tables(2)
    xq, wq = o.gauss(3)

    def load_vector(t):
        F = np.zeros((3, ndu, ndu, ndu))
        W = h ** 3 * np.einsum("i,j,k->ijk", wq, wq, wq)
        for cz in range(n):
            for cy in range(n):
                for cx in range(n):
                    X, Y, Zc = h * (cx + xq)[None, None, :], h * (cy + xq)[None, :, None], h * (cz + xq)[:, None, None]
                    f = stokes3d_force(X, Y, Zc, t, nu)
                    for c in range(3):
                        F[c, 2 * cz:2 * cz + 3, 2 * cy:2 * cy + 3, 2 * cx:2 * cx + 3] += np.einsum("zyx,za,yb,xc->abc", W * f[c], Su, Su, Su)
        return F.reshape(3 * Nu)[free]

    tq_int = o.gauss_radau_right(k + 1) if ttype == o.DG else o.gauss_lobatto(k + 1)
    et, ewt = o.gauss(k + 1)
    Ltime, _ = lagrange_eval(tq_int, et)
    eu, ewu = o.gauss(3)
    ep, ewp = o.gauss(2)
    Eu, dEu = lagrange_eval(o.gauss_lobatto(3), eu)
    Ep, _ = lagrange_eval(o.gauss_lobatto(2), ep)

    def errors_u(uf, t):
        U = np.zeros(3 * Nu)
        U[free] = uf
        U = U.reshape(3, ndu, ndu, ndu)
        l2 = h1 = l8 = 0.0
        W = h ** 3 * np.einsum("i,j,k->ijk", ewu, ewu, ewu)
        for cz in range(n):
            for cy in range(n):
                for cx in range(n):
                    X, Y, Zc = h * (cx + eu)[None, None, :], h * (cy + eu)[None, :, None], h * (cz + eu)[:, None, None]
                    ue, ge = stokes3d_exact_u(X, Y, Zc, t), stokes3d_exact_grad_u(X, Y, Zc, t)
                    for c in range(3):
                        loc = U[c, 2 * cz:2 * cz + 3, 2 * cy:2 * cy + 3, 2 * cx:2 * cx + 3]
                        uh = np.einsum("ac,bd,ef,cdf->abe", Eu, Eu, Eu, loc)
                        ux = np.einsum("ac,bd,ef,cdf->abe", Eu, Eu, dEu, loc) / h
                        uy = np.einsum("ac,bd,ef,cdf->abe", Eu, dEu, Eu, loc) / h
                        uz = np.einsum("ac,bd,ef,cdf->abe", dEu, Eu, Eu, loc) / h
                        l2 += np.sum(W * (uh - ue[c]) ** 2)
                        h1 += np.sum(W * ((ux - ge[c][0]) ** 2 + (uy - ge[c][1]) ** 2 + (uz - ge[c][2]) ** 2))
                        l8 = max(l8, np.abs(uh - ue[c]).max())
        return l2, l8, h1

    def errors_p(pf, t):
        P = pf.reshape(n, n, n, 4) if dg_pressure else pf.reshape(ndp, ndp, ndp)
        l2 = 0.0
        W = h ** 3 * np.einsum("i,j,k->ijk", ewp, ewp, ewp)
        lg = np.sqrt(3.0) * (2 * ep - 1)  # deal.II's Legendre basis of FE_DGP(1): 1, l(xi), l(eta), l(zeta)
        for cz in range(n):
            for cy in range(n):
                for cx in range(n):
                    X, Y, Zc = h * (cx + ep)[None, None, :], h * (cy + ep)[None, :, None], h * (cz + ep)[:, None, None]
                    if dg_pressure:
                        c = P[cz, cy, cx]
                        ph = c[0] + c[1] * lg[None, None, :] + c[2] * lg[None, :, None] + c[3] * lg[:, None, None]
                    else:
                        ph = np.einsum("ac,bd,ef,cdf->abe", Ep, Ep, Ep, P[cz:cz + 2, cy:cy + 2, cx:cx + 2])
                    l2 += np.sum(W * (ph - stokes3d_exact_p(X, Y, Zc, t)) ** 2)
        return l2

    # mean value: (1, psi_j) p_j / |Omega|; FE_Q(1): the exact mass applied to 1 (trapezoid weights per direction); FE_DGP(1): psi_0 = 1
    # and the other functions have zero mean on a box.  The shift subtracts the mean times the coefficients of the constant 1.
    if dg_pressure:
        mean_w = np.zeros(Np)
        mean_w[0::4] = h ** 3
        one_p = np.zeros(Np)
        one_p[0::4] = 1.0
    else:
        w1 = np.full(ndp, h)
        w1[0] = w1[-1] = h / 2
        mean_w = np.einsum("i,j,k->ijk", w1, w1, w1).ravel()
        one_p = np.ones(Np)
    prev_u, prev_p = np.zeros(NU), np.zeros(NP)
    time = 0.0
    acc_l2 = acc_h1 = acc_p = 0.0
    acc_l8 = -1.0
    while time < 1.0 - 1e-12:
        rhs = np.zeros(N)
        KSu = KS_uu @ prev_u + Bt @ prev_p
        KSp = Bm @ prev_u
        Mu = MM @ prev_u
        for a in range(nt):
            rhs[ub(a)] = rKu[a] * KSu + rM[a] * Mu
            rhs[pb(a)] = rKp[a] * KSp
        for j, xi in enumerate(tq_int):
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
        xp = [sol[pb(a)] - np.dot(mean_w, sol[pb(a)]) * one_p for a in range(nt)]
        for q in range(k + 1):
            if ttype == o.DG:
                uf = sum(Ltime[q, i] * xu[i] for i in range(nt))
                pf = sum(Ltime[q, i] * xp[i] for i in range(nt))
            else:
                uf = Ltime[q, 0] * prev_u + sum(Ltime[q, i] * xu[i - 1] for i in range(1, k + 1))
                pf = Ltime[q, 0] * prev_p + sum(Ltime[q, i] * xp[i - 1] for i in range(1, k + 1))
            t = time + tau * et[q]
            l2, l8, h1 = errors_u(uf, t)
            acc_l2 += tau * ewt[q] * l2
            acc_h1 += tau * ewt[q] * h1
            acc_l8 = max(acc_l8, l8)
            acc_p += tau * ewt[q] * errors_p(pf, t)
        prev_u, prev_p = xu[-1], xp[-1]
        time += tau
    return acc_l8, np.sqrt(acc_l2), np.sqrt(acc_h1), np.sqrt(acc_p)
