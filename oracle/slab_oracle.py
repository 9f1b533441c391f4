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
