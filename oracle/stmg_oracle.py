"""CPU restatement (numpy / scipy.sparse) of the reference's space-time multigrid: the space transfers
deal.II's MGTwoLevelTransfer provides to include/stmg.h:38-110, the time transfers of stmg.h:113-247, the level
bookkeeping of stmg.h:460-501 / fe_time.h:412-443 and the V-cycle the reference assembles in GMG::reinit
(stmg.h:1190-1327) from Multigrid, MGSmootherPrecondition, PreconditionRelaxation and MGCoarseGridApplySmoother.
TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product path.

Structure follows deal.II, not the product: the space transfer is assembled CELL BY CELL (per fine cell: the
embedding of its parent's - or its own coarser-degree - shape functions, weighted by the inverse valence of the fine
DoF, added into the global matrix, constrained rows and columns dropped), whereas the product applies Kronecker
factors of banded 1D matrices.

Parity status: the time-transfer matrices are pinned by the reference's tests/transfer_02.output
(tests/test_time_transfers.py) and the level schedule by the known answers of tests/tp04.cc
(tests/golden/mg_sequence_cases.json); the space transfers and the V-cycle are unpinned by reference-held numbers (the
reference's transfer_01.output holds iteration counts of 2D runs only) and are checked by properties: constants and
polynomials of the coarse degree are reproduced, restriction = transpose, Galerkin identity P^T M_f P = M_c."""
import numpy as np
import scipy.sparse as sp

from . import oracle as _o


def _lagrange(nodes, x):
    nodes = np.asarray(nodes, float)
    out = np.ones(len(nodes))
    for a in range(len(nodes)):
        for m in range(len(nodes)):
            if m != a:
                out[a] *= (x - nodes[m]) / (nodes[a] - nodes[m])
    return out


def constrained_mask(p, nc, dirichlet_mask):
    """bool [N]: DoFs on faces carrying the zero boundary condition (bit 2d: lower, 2d+1: upper face of direction d)"""
    nd = [p * c + 1 for c in nc]
    con = np.zeros(nd[::-1], bool)
    for d in range(3):
        idx = [slice(None)] * 3
        if dirichlet_mask >> (2 * d) & 1:
            idx[2 - d] = 0
            con[tuple(idx)] = True
        idx = [slice(None)] * 3
        if dirichlet_mask >> (2 * d + 1) & 1:
            idx[2 - d] = -1
            con[tuple(idx)] = True
    return con.ravel()


def _cell_dofs(p, nd, cx, cy, cz):
    i = np.arange(p + 1)
    return ((cx * p + i)[None, None, :] + nd[0] * ((cy * p + i)[None, :, None] + nd[1] * (cz * p + i)[:, None, None])).ravel()


def space_prolongation(p_f, nc_f, mask_f, p_c, nc_c, mask_c):
    """P [N_f x N_c] as deal.II's MGTwoLevelTransfer applies it (prolongate_and_add), cell by cell"""
    nd_f = [p_f * c + 1 for c in nc_f]
    nd_c = [p_c * c + 1 for c in nc_c]
    r = [nc_f[d] // nc_c[d] for d in range(3)]
    assert all(nc_f[d] == r[d] * nc_c[d] and r[d] in (1, 2) for d in range(3))
    gf, gc = _o.gauss_lobatto(p_f + 1), _o.gauss_lobatto(p_c + 1)
    # 1D embedding of the coarse cell's basis into child `s` of `r` children
    local = {(rr, s): np.array([_lagrange(gc, (s + x) / rr) for x in gf]) for rr in (1, 2) for s in range(rr)}
    Nf, Nc = int(np.prod(nd_f)), int(np.prod(nd_c))
    valence = np.zeros(Nf)
    for cz in range(nc_f[2]):
        for cy in range(nc_f[1]):
            for cx in range(nc_f[0]):
                valence[_cell_dofs(p_f, nd_f, cx, cy, cz)] += 1
    rows, cols, vals = [], [], []
    for cz in range(nc_f[2]):
        for cy in range(nc_f[1]):
            for cx in range(nc_f[0]):
                f = _cell_dofs(p_f, nd_f, cx, cy, cz)
                c = _cell_dofs(p_c, nd_c, cx // r[0], cy // r[1], cz // r[2])
                L = np.kron(local[r[2], cz % r[2]], np.kron(local[r[1], cy % r[1]], local[r[0], cx % r[0]]))
                L = L / valence[f][:, None]
                rows.append(np.repeat(f, len(c)))
                cols.append(np.tile(c, len(f)))
                vals.append(L.ravel())
    P = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(Nf, Nc)).tocsr()
    keep_f = sp.diags((~constrained_mask(p_f, nc_f, mask_f)).astype(float))
    keep_c = sp.diags((~constrained_mask(p_c, nc_c, mask_c)).astype(float))
    return (keep_f @ P @ keep_c).tocsr()


def space_interpolation(p_f, nc_f, mask_f, p_c, nc_c, mask_c):
    """I [N_c x N_f]: the fine function at the coarse nodes (MGTwoLevelTransfer::interpolate)"""
    nd_f = [p_f * c + 1 for c in nc_f]
    nd_c = [p_c * c + 1 for c in nc_c]
    gf, gc = _o.gauss_lobatto(p_f + 1), _o.gauss_lobatto(p_c + 1)
    lines = []
    for d in range(3):
        r = nc_f[d] // nc_c[d]
        A = np.zeros((nd_c[d], nd_f[d]))
        for c in range(nd_c[d]):
            cell, j = (nc_c[d] - 1, p_c) if c == p_c * nc_c[d] else divmod(c, p_c)
            t = gc[j] * r
            sub = min(int(t), r - 1)
            A[c, (cell * r + sub) * p_f:(cell * r + sub) * p_f + p_f + 1] = _lagrange(gf, t - sub)
        lines.append(sp.csr_matrix(A))
    full = sp.kron(lines[2], sp.kron(lines[1], lines[0]))
    keep_f = sp.diags((~constrained_mask(p_f, nc_f, mask_f)).astype(float))
    keep_c = sp.diags((~constrained_mask(p_c, nc_c, mask_c)).astype(float))
    return (keep_c @ full @ keep_f).tocsr()


def blk_dofs(ttype, r):
    return r + 1 if ttype == 1 else r


def level_structure(ttype, n_timesteps_at_once, mg_type_level, poly_time_sequence):
    """stmg.h:460-501 and fe_time.h:412-443: per level (coarsest first) the temporal degree, the number of time steps
    per slab and the factor on the time step size"""
    n_levels = len(mg_type_level) + 1
    out = [None] * n_levels
    deg = list(poly_time_sequence)
    pi, n, scale = len(deg) - 1, n_timesteps_at_once, 1.0
    out[-1] = (deg[pi], n, scale)
    for i in range(n_levels - 2, -1, -1):
        m = mg_type_level[i]
        if m == "k":
            pi -= 1
        elif m == "t":
            n //= 2
            scale *= 2
        out[i] = (deg[pi], n, scale)
    assert pi == 0
    return out


def time_transfer(ttype, kind, r_hi, r_lo, n_hi, restrict_is_transpose_prolongate=True):
    """stmg.h:166-215: (prolongation, restriction) matrices of a 'k' or 't' transfer"""
    if kind == "k":
        P = _o.time_projection(ttype, r_lo, r_hi, n_hi)
        down = _o.time_projection(ttype, r_hi, r_lo, n_hi)
    else:
        P = _o.time_prolongation(ttype, r_hi, n_hi)
        down = _o.time_restriction(ttype, r_hi, n_hi)
    return P, (P.T.copy() if restrict_is_transpose_prolongate else down)


def power_iteration(A, Pinv, nb, n, n_iterations=20):
    """largest eigenvalue of P^-1 A by deal.II's power iteration from the vector (i mod 11) - mean on every block"""
    g = np.arange(n) % 11
    g = g - g.mean()
    v = np.tile(g, nb).astype(float)
    v /= np.linalg.norm(v)
    lam = 0.0
    for _ in range(n_iterations):
        w = Pinv(A @ v)
        lam = v @ w
        if not np.linalg.norm(w) > 0:
            break
        v = w / np.linalg.norm(w)
    return abs(lam)


def chebyshev_interval(lam, smoothing_range=1.0):
    """(alpha, beta) as deal.II derives them from the power iteration: upper estimate 1.2 lambda, lower estimate 1"""
    beta = 1.2 * lam
    alpha = beta / smoothing_range if smoothing_range > 1 else min(0.9 * beta, 1.0)
    return alpha, beta


def power_iteration_relaxation(A, Pinv, nb, n, n_iterations=20, smoothing_range=1.0):
    """deal.II PreconditionRelaxation with relaxation = 0: 2 / (alpha + beta)"""
    lam = power_iteration(A, Pinv, nb, n, n_iterations)
    if not lam > 0:  # a level without free DoFs
        return 1.0
    alpha, beta = chebyshev_interval(lam, smoothing_range)
    return 2.0 / (alpha + beta)


class Multigrid:
    """One V-cycle as PreconditionMG::vmult runs it.  levels[l] = dict(A = matrix, smoother = callable r -> P^-1 r or
    None for the identity, omega, n_iterations); transfers[l] = (P, R) between level l - 1 and l (matrices acting on the
    flattened block vectors)."""

    def __init__(self, levels, transfers, variable=True, steps=1, coarse_gmres=None):
        """coarse_gmres = (maxiter, abstol): the reference's coarseGridSmootherType != "Smoother" (stmg.h:1240-1308):
        MGCoarseGridIterativeSolver around SolverGMRES(IterationNumberControl(maxiter, abstol), basis maxiter), left-preconditioned
        by the coarsest level's relaxation preconditioner, from the zero vector."""
        self.levels, self.transfers, self.variable, self.steps = levels, transfers, variable, steps
        self.coarse_gmres = coarse_gmres

    def _coarse_gmres(self, b):
        """The GMRES iterate in its defining form: x_k minimises || P^-1 (b - A x) || over the Krylov space
        K_k(P^-1 A, P^-1 b) (dense least squares on an orthonormalised basis, no Arnoldi recurrence), k = the first step at
        which the preconditioned residual is below abstol, else maxiter."""
        maxiter, abstol = self.coarse_gmres
        A = self.levels[0]["A"]
        M = lambda v: self._precondition(0, v)  # noqa: E731
        r0 = M(b)
        if np.linalg.norm(r0) <= abstol:
            return np.zeros_like(b)
        K = [r0 / np.linalg.norm(r0)]
        x = np.zeros_like(b)
        for k in range(1, maxiter + 1):
            Q, _ = np.linalg.qr(np.stack(K, axis=1))
            MAQ = np.stack([M(A @ Q[:, i]) for i in range(Q.shape[1])], axis=1)
            y, *_ = np.linalg.lstsq(MAQ, r0, rcond=None)
            x = Q @ y
            if np.linalg.norm(r0 - MAQ @ y) <= abstol or k == maxiter:
                break
            w = M(A @ K[-1])
            K.append(w / np.linalg.norm(w))
        return x

    def _precondition(self, l, r):
        lv = self.levels[l]
        if lv["smoother"] is None:  # PreconditionIdentity
            return r.copy()
        if lv.get("chebyshev"):  # PreconditionChebyshev: `degree` applications of P^-1 on [alpha, beta], from 0
            alpha, beta = lv["chebyshev"]
            theta, delta = 0.5 * (beta + alpha), 0.5 * (beta - alpha)
            x_old, x = np.zeros_like(r), lv["smoother"](r) / theta
            sigma = theta / delta
            rho = 1.0 / sigma
            for _ in range(1, lv["n_iterations"]):
                rho_new = 1.0 / (2.0 * sigma - rho)
                x, x_old = x + rho_new * rho * (x - x_old) + 2.0 * rho_new / delta * lv["smoother"](r - lv["A"] @ x), x
                rho = rho_new
            return x
        # PreconditionRelaxation: n_iterations sweeps from 0
        x = lv["omega"] * lv["smoother"](r)
        for _ in range(1, lv["n_iterations"]):
            x = x + lv["omega"] * lv["smoother"](r - lv["A"] @ x)
        return x

    def _smooth(self, l, u, rhs, from_zero):
        steps = self.steps * (2 ** (len(self.levels) - 1 - l) if self.variable else 1)
        i = 0
        if from_zero:
            u = self._precondition(l, rhs)
            i = 1
        for _ in range(i, steps):
            u = u + self._precondition(l, rhs - self.levels[l]["A"] @ u)
        return u

    def _v(self, l, defect):
        if l == 0:
            return self._coarse_gmres(defect) if self.coarse_gmres else self._smooth(0, None, defect, True)
        u = self._smooth(l, None, defect, True)
        t = defect - self.levels[l]["A"] @ u
        P, R = self.transfers[l]
        u = u + P @ self._v(l - 1, R @ t)
        return self._smooth(l, u, defect, False)

    def vmult(self, src):
        return self._v(len(self.levels) - 1, np.asarray(src, float).ravel())
