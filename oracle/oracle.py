"""ctypes binding of the CPU ORACLE (oracle/stfem_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product (dealii-stfem_amd) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libstfem_oracle.so")


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("stfem_oracle.c", "stfem_oracle_stokes.c", "stfem_cpu_baseline.c", "stfem_oracle.h")]
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libstfem_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB


_lib = None
_dp = C.POINTER(C.c_double)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        L.stfo_create.restype = C.c_void_p
        L.stfo_create.argtypes = [C.c_int, C.POINTER(C.c_int), _dp, C.c_int]
        L.stfo_destroy.argtypes = [C.c_void_p]
        L.stfo_n_dofs.restype = C.c_long
        L.stfo_n_dofs.argtypes = [C.c_void_p]
        L.stfo_n_cells.restype = C.c_long
        L.stfo_n_cells.argtypes = [C.c_void_p]
        L.stfo_n_q.argtypes = [C.c_void_p]
        L.stfo_set_threads.argtypes = [C.c_int]
        L.stfo_set_coefficient.argtypes = [C.c_void_p, C.c_int, _dp]
        L.stfo_quadrature_points.argtypes = [C.c_void_p, _dp]
        L.stfo_coefficient_values.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double,
                                              C.c_double, C.POINTER(C.c_int), _dp, _dp, _dp]
        L.stfo_space_vmult.argtypes = [C.c_void_p, C.c_double, C.c_double, _dp, _dp]
        L.stfo_st_vmult.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int,
                                    C.POINTER(_dp), C.POINTER(_dp)]
        L.stfo_diagonal.argtypes = [C.c_void_p, C.c_double, C.c_double, _dp]
        L.stfo_dense.argtypes = [C.c_void_p, C.c_double, C.c_double, _dp]
        L.stfo_gauss.argtypes = [C.c_int, _dp, _dp]
        L.stfo_gauss_lobatto.argtypes = [C.c_int, _dp]
        L.stfo_gauss_radau_right.argtypes = [C.c_int, _dp]
        L.stfo_shape_tables.argtypes = [C.c_int, C.c_int, _dp, _dp]
        L.stfo_cg_weights.argtypes = [C.c_int, _dp, _dp]
        L.stfo_dg_weights.argtypes = [C.c_int, _dp, _dp, _dp]
        L.stfo_time_weights.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, _dp, _dp, _dp, _dp]
        L.stfo_time_weights_wave.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int,
                                             _dp, _dp, _dp, _dp, _dp]
        L.stfo_stokes_n_velocity.restype = C.c_long
        L.stfo_stokes_n_velocity.argtypes = [C.POINTER(C.c_int), C.c_int]
        L.stfo_stokes_n_pressure.restype = C.c_long
        L.stfo_stokes_n_pressure.argtypes = [C.POINTER(C.c_int), C.c_int]
        L.stfo_stokes_apply.argtypes = [C.POINTER(C.c_int), _dp, C.c_int, C.c_int, C.c_double,
                                        C.c_double, C.c_double, _dp, _dp, _dp, _dp, C.c_int]
        L.stfo_stokes_set_pressure_space.argtypes = [C.c_int]
        L.stfo_stokes_n_pressure_space.restype = C.c_long
        L.stfo_stokes_n_pressure_space.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int]
        L.stfo_stokes_n_face_points.restype = C.c_long
        L.stfo_stokes_n_face_points.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int]
        L.stfo_stokes_face_points.argtypes = [C.POINTER(C.c_int), _dp, C.c_int, C.c_int, _dp]
        L.stfo_stokes_boundary_apply.argtypes = [C.POINTER(C.c_int), _dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                                 C.c_double, _dp, _dp, _dp, _dp]
        L.stfo_stokes_nitsche_rhs.argtypes = [C.POINTER(C.c_int), _dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                              _dp, _dp, _dp]
        L.stfb_st_vmult.argtypes = [C.c_int, C.POINTER(C.c_int), _dp, _dp, C.c_int, C.c_int, _dp, _dp,
                                    C.POINTER(_dp), C.POINTER(_dp), _dp, C.c_int]
        # default thread count: the visible CPUs, but never more than 16 (a GPU box advertises 256
        # logical CPUs and grants far fewer; oversubscribed libgomp threads spin)
        L.stfo_set_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp)


def gauss(n):
    x = np.zeros(n); w = np.zeros(n)
    lib().stfo_gauss(n, _p(x), _p(w))
    return x, w


def gauss_lobatto(n):
    x = np.zeros(n)
    lib().stfo_gauss_lobatto(n, _p(x))
    return x


def gauss_radau_right(n):
    x = np.zeros(n)
    lib().stfo_gauss_radau_right(n, _p(x))
    return x


def shape_tables(p, nq=None):
    nq = nq or p + 1
    S = np.zeros((nq, p + 1)); D = np.zeros((nq, p + 1))
    lib().stfo_shape_tables(p, nq, _p(S), _p(D))
    return S, D


CGP, DG = 0, 1


def cg_weights(r):
    M = np.zeros((r, r + 1)); D = np.zeros((r, r + 1))
    assert lib().stfo_cg_weights(r, _p(M), _p(D)) == 0
    return M, D


def dg_weights(r):
    M = np.zeros((r + 1, r + 1)); D = np.zeros((r + 1, r + 1)); j = np.zeros((r + 1, 1))
    assert lib().stfo_dg_weights(r, _p(M), _p(D), _p(j)) == 0
    return M, D, j


def time_weights(ttype, r, tau=1.0, nsteps=1):
    nb = (r if ttype == CGP else r + 1) * nsteps
    A = np.zeros((nb, nb)); B = np.zeros((nb, nb)); G = np.zeros((nb, 1)); Z = np.zeros((nb, 1))
    assert lib().stfo_time_weights(ttype, r, tau, nsteps, _p(A), _p(B), _p(G), _p(Z)) == nb
    return A, B, G, Z


def time_weights_wave(ttype, r, tau=1.0, nsteps=1):
    nb = (r if ttype == CGP else r + 1) * nsteps
    A = np.zeros((nb, nb)); B = np.zeros((nb, nb))
    v = [np.zeros((nb, 1)) for _ in range(3)]
    assert lib().stfo_time_weights_wave(ttype, r, tau, nsteps, _p(A), _p(B),
                                        _p(v[0]), _p(v[1]), _p(v[2])) == nb
    return A, B, v[0], v[1], v[2]


class Oracle:
    """CPU restatement of MatrixFreeOperator + SystemMatrix on a structured hex mesh."""

    def __init__(self, p, ncell, vertices, dirichlet_mask=63):
        self.p = p
        self.ncell = tuple(int(v) for v in ncell)
        nc = (C.c_int * 3)(*self.ncell)
        v = np.ascontiguousarray(vertices, dtype=np.float64)
        assert v.size == 3 * np.prod([n + 1 for n in self.ncell])
        self._h = lib().stfo_create(p, nc, _p(v), dirichlet_mask)
        assert self._h
        self.n_dofs = lib().stfo_n_dofs(self._h)
        self.n_cells = lib().stfo_n_cells(self._h)
        self.nq = lib().stfo_n_q(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().stfo_destroy(self._h)
            self._h = None

    def set_coefficient(self, which, coef):
        if coef is None:
            lib().stfo_set_coefficient(self._h, which, None)
        else:
            c = np.ascontiguousarray(coef, dtype=np.float64)
            assert c.size == self.n_cells * self.nq ** 3
            lib().stfo_set_coefficient(self._h, which, _p(c))

    def quadrature_points(self):
        out = np.zeros((self.n_cells, self.nq ** 3, 3))
        lib().stfo_quadrature_points(self._h, _p(out))
        return out

    def coefficient_values(self, c1=1.0, c2=9.0, c3=16.0, distort=0.0, subdivisions=(1, 1, 1),
                           lower=(0, 0, 0), upper=(1, 1, 1)):
        out = np.zeros((self.n_cells, self.nq ** 3))
        sub = (C.c_int * 3)(*subdivisions)
        lo = np.array(lower, dtype=np.float64); up = np.array(upper, dtype=np.float64)
        lib().stfo_coefficient_values(self._h, c1, c2, c3, distort, sub, _p(lo), _p(up), _p(out))
        return out

    def space_vmult(self, src, mass=0.0, laplace=0.0):
        src = np.ascontiguousarray(src, dtype=np.float64)
        dst = np.zeros(self.n_dofs)
        lib().stfo_space_vmult(self._h, mass, laplace, _p(dst), _p(src))
        return dst

    def st_vmult(self, alpha, beta, src, transpose=False, dst=None):
        """src: (n_src_blocks, n_dofs). Returns (n_dst_blocks, n_dofs); adds into dst if given."""
        alpha = np.ascontiguousarray(alpha, dtype=np.float64)
        beta = np.ascontiguousarray(beta, dtype=np.float64)
        nrows, ncols = alpha.shape
        src = np.ascontiguousarray(src, dtype=np.float64)
        nsrc, ndst = (nrows, ncols) if transpose else (ncols, nrows)
        assert src.shape == (nsrc, self.n_dofs)
        add = dst is not None
        out = np.ascontiguousarray(dst, dtype=np.float64).copy() if add else \
            np.zeros((ndst, self.n_dofs))
        sp = (_dp * nsrc)(*[_p(src[i]) for i in range(nsrc)])
        dp = (_dp * ndst)(*[_p(out[i]) for i in range(ndst)])
        lib().stfo_st_vmult(self._h, nrows, ncols, _p(alpha), _p(beta), int(transpose), int(add),
                            dp, sp)
        return out

    def diagonal(self, mass=0.0, laplace=0.0):
        d = np.zeros(self.n_dofs)
        lib().stfo_diagonal(self._h, mass, laplace, _p(d))
        return d

    def dense(self, mass=0.0, laplace=0.0):
        A = np.zeros((self.n_dofs, self.n_dofs))
        lib().stfo_dense(self._h, mass, laplace, _p(A))
        return A


# ---- Stokes two-field operator (stfem_oracle_stokes.c)

def stokes_block_index(nt, it, v, d, n_variables=2, variable_major=True):
    """BlockSlice::index (reference include/fe_time.h:956-967)"""
    if variable_major:
        return it * (n_variables * nt) + v * nt + d
    return it * (n_variables * nt) + d * n_variables + v


class StokesOracle:
    """StokesMatrixFreeOperator (cell loop) + vector mass + SystemMatrixStokes::vmult, restated in
    the reference's structure (operators.h:825-867: K.vmult, scatter with Alpha, M.vmult, scatter
    with Beta, one source time dof after the other)."""

    def __init__(self, ncell, vertices, dirichlet_mask, viscosity, pu=2, weak_mask=0, penalty1=20.0, penalty2=10.0, dg_pressure=False):
        self.nc = (C.c_int * 3)(*ncell)
        self.vertices = np.ascontiguousarray(vertices, dtype=np.float64).reshape(-1)
        self.mask, self.nu, self.pu = int(dirichlet_mask), float(viscosity), pu
        # weak (Nitsche) boundary faces, operators.h:1206-1211, 1220-1221: gamma1 = nu penalty1, gamma2 = penalty2
        self.weak, self.penalty1, self.penalty2 = int(weak_mask), float(penalty1), float(penalty2)
        self.pspace = 1 if dg_pressure else 0  # FE_DGP(pu - 1) instead of FE_Q(pu - 1): tests/tp_03stokes.cc:83-86
        self.n_u = lib().stfo_stokes_n_velocity(self.nc, pu)
        self.n_p = lib().stfo_stokes_n_pressure_space(self.nc, pu, self.pspace)

    def apply(self, U, P, wK=1.0, wM=0.0):
        lib().stfo_stokes_set_pressure_space(self.pspace)
        U = np.ascontiguousarray(U, dtype=np.float64).reshape(3 * self.n_u)
        P = np.ascontiguousarray(P, dtype=np.float64).reshape(self.n_p)
        ou = np.zeros(3 * self.n_u); op = np.zeros(self.n_p)
        rc = lib().stfo_stokes_apply(self.nc, _p(self.vertices), self.pu, self.mask, self.nu, wK, wM,
                                     _p(U), _p(P), _p(ou), _p(op), 0)
        assert rc == 0
        if self.weak and wK != 0.0:  # LoopType::Full: the boundary-face loop of the same vmult
            rc = lib().stfo_stokes_boundary_apply(self.nc, _p(self.vertices), self.pu, self.mask, self.weak, self.nu, self.penalty1,
                                                  self.penalty2, wK, _p(U), _p(P), _p(ou), _p(op))
            assert rc == 0
        return ou.reshape(3, self.n_u), op

    def face_points(self):
        n = lib().stfo_stokes_n_face_points(self.nc, self.pu, self.weak)
        out = np.zeros((n, 3))
        assert lib().stfo_stokes_face_points(self.nc, _p(self.vertices), self.pu, self.weak, _p(out)) == 0
        return out

    def nitsche_rhs(self, g_at_face_points):
        """StokesNitscheMatrixFreeOperator::vmult (operators.h:1833-1849, 1898-1940) for the Dirichlet data at face_points()"""
        lib().stfo_stokes_set_pressure_space(self.pspace)
        g = np.ascontiguousarray(g_at_face_points, dtype=np.float64)
        ou = np.zeros(3 * self.n_u); op = np.zeros(self.n_p)
        assert lib().stfo_stokes_nitsche_rhs(self.nc, _p(self.vertices), self.pu, self.mask, self.weak, self.nu, self.penalty1,
                                             self.penalty2, _p(g), _p(ou), _p(op)) == 0
        return ou.reshape(3, self.n_u), op

    def st_vmult(self, Alpha, Beta, n_timesteps, n_timedofs, blocks, variable_major=True):
        """blocks: list of arrays in BlockSlice order (velocity blocks 3*n_u, pressure n_p)."""
        nt = n_timedofs
        idx = lambda it, v, d: stokes_block_index(nt, it, v, d, 2, variable_major)  # noqa: E731
        eps10 = 10 * np.finfo(np.float64).eps
        dst = [np.zeros_like(np.asarray(b, dtype=np.float64).reshape(-1)) for b in blocks]
        for it in range(n_timesteps):
            for d in range(nt):
                u = np.asarray(blocks[idx(it, 0, d)]).reshape(3, self.n_u)
                p = np.asarray(blocks[idx(it, 1, d)]).reshape(self.n_p)
                tu, tp = self.apply(u, p, 1.0, 0.0)          # K.vmult(tmp, tmp_src)
                i = idx(it, 0, d)
                for jt in range(n_timesteps):
                    for jd in range(nt):
                        for jv, t in ((0, tu.reshape(-1)), (1, tp)):
                            j = idx(jt, jv, jd)
                            if abs(Alpha[j, i]) > eps10:      # internal::scatter, operators.h:91-110
                                dst[j] += Alpha[j, i] * t
                mu, _ = self.apply(u, np.zeros(self.n_p), 0.0, 1.0)  # M.vmult(tmp.block(0), ...)
                for jt in range(n_timesteps):
                    for jd in range(nt):
                        j = idx(jt, 0, jd)
                        if abs(Beta[j, i]) > eps10:
                            dst[j] += Beta[j, i] * mu.reshape(-1)
        return dst

    def st_Tvmult(self, Alpha, Beta, n_timesteps, n_timedofs, blocks, variable_major=True):
        """SystemMatrixStokes::Tvmult as the reference has it (operators.h:708-745): the nine-argument scatter overload
        (operators.h:111-123) reads  j = index(it, v, id), i = index(jt, v, jd), so every source time dof (it, id) only feeds the
        destination blocks of ITS OWN time dof, weighted with entries of row j summed over (jt, jd) - not a transpose."""
        nt = n_timedofs
        idx = lambda it, v, d: stokes_block_index(nt, it, v, d, 2, variable_major)  # noqa: E731
        eps10 = 10 * np.finfo(np.float64).eps
        dst = [np.zeros_like(np.asarray(b, dtype=np.float64).reshape(-1)) for b in blocks]
        for it in range(n_timesteps):
            for d in range(nt):
                u = np.asarray(blocks[idx(it, 0, d)]).reshape(3, self.n_u)
                p = np.asarray(blocks[idx(it, 1, d)]).reshape(self.n_p)
                tu, tp = self.apply(u, p, 1.0, 0.0)
                for jt in range(n_timesteps):
                    for jd in range(nt):
                        for v, t in ((0, tu.reshape(-1)), (1, tp)):
                            j, i = idx(it, v, d), idx(jt, v, jd)
                            if abs(Alpha[j, i]) > eps10:
                                dst[j] += Alpha[j, i] * t
                mu, _ = self.apply(u, np.zeros(self.n_p), 0.0, 1.0)
                for jt in range(n_timesteps):
                    for jd in range(nt):
                        j, i = idx(it, 0, d), idx(jt, 0, jd)
                        if abs(Beta[j, i]) > eps10:
                            dst[j] += Beta[j, i] * mu.reshape(-1)
        return dst


# ---- CPU baseline (stfem_cpu_baseline.c): the reference's structure the way MatrixFree runs it

class CpuBaseline:
    """SystemMatrix::vmult on a Cartesian box: 2 n_blocks spatial cell loops + axpys
    (include/operators.h:536-559), Cartesian-compressed geometry, SIMD across cells, OpenMP."""

    def __init__(self, p, ncell, lower=(0, 0, 0), upper=(1, 1, 1), dirichlet_mask=63, threads=0):
        self.p = p
        self.nc = (C.c_int * 3)(*ncell)
        self.lower = np.array(lower, dtype=np.float64)
        self.upper = np.array(upper, dtype=np.float64)
        self.mask = dirichlet_mask
        self.threads = threads or max(1, min(64, len(os.sched_getaffinity(0))))
        self.n_dofs = int(np.prod([p * n + 1 for n in ncell]))
        self._tmp = np.empty(self.n_dofs)

    def st_vmult(self, Alpha, Beta, X, Y=None):
        nb = Alpha.shape[0]
        assert Alpha.shape == (nb, nb) and X.shape == (nb, self.n_dofs)
        A = np.ascontiguousarray(Alpha, dtype=np.float64)
        B = np.ascontiguousarray(Beta, dtype=np.float64)
        if Y is None:
            Y = np.empty_like(X)
        src = (_dp * nb)(*[_p(X[b]) for b in range(nb)])
        dst = (_dp * nb)(*[_p(Y[b]) for b in range(nb)])
        rc = lib().stfb_st_vmult(self.p, self.nc, _p(self.lower), _p(self.upper), self.mask, nb, _p(A), _p(B),
                                 src, dst, _p(self._tmp), self.threads)
        assert rc == 0
        return Y


# ---- time-multigrid transfer matrices (reference include/fe_time.h:749-898): restated with the deal.II pieces they
# call spelled out - FiniteElement::get_prolongation_matrix / get_restriction_matrix of the 1D Lagrange elements on
# the Gauss-Lobatto (cG) / right Gauss-Radau (dG) points, and FETools::get_projection_matrix (L2 projection on the
# reference cell).  Pinned by the reference's tests/transfer_02.output (tests/test_time_transfers.py).
def _lagrange(nodes, x):
    nodes = np.asarray(nodes, float)
    x = np.atleast_1d(np.asarray(x, float))
    L = np.ones((len(x), len(nodes)))
    for a in range(len(nodes)):
        for m in range(len(nodes)):
            if m != a:
                L[:, a] *= (x - nodes[m]) / (nodes[a] - nodes[m])
    return L


def _time_nodes(ttype, r):
    return np.asarray(gauss_lobatto(r + 1) if ttype == CGP else gauss_radau_right(r + 1), float)


def time_prolongation(ttype, r, nsteps=2):
    """get_time_prolongation_matrix (fe_time.h:805-849): two fine time steps <- one coarse step of twice the length"""
    x = _time_nodes(ttype, r)
    left, right = _lagrange(x, x / 2), _lagrange(x, (x + 1) / 2)  # parent basis at the children's support points
    if ttype == CGP:  # the dof at the left end of a step belongs to the previous step
        P = np.vstack([left[1:, 1:], right[1:, 1:]])
    else:
        P = np.vstack([left, right])
    n = P.shape[1]
    out = np.zeros((n * nsteps, n * nsteps // 2))
    for it in range(nsteps // 2):
        out[2 * n * it:2 * n * (it + 1), n * it:n * (it + 1)] = P
    return out


def time_restriction(ttype, r, nsteps=2):
    """get_time_restriction_matrix (fe_time.h:851-898).  cG (FE_Q): the parent's value at its support point, read
    from the child the point lies in; dG (FE_DGQArbitraryNodes): L2 projection of the two children."""
    x = _time_nodes(ttype, r)
    if ttype == CGP:
        eps = 1e-12
        left = np.where((x <= 0.5 + eps)[:, None], _lagrange(x, np.clip(2 * x, 0, 1)), 0.0)
        right = np.where((x >= 0.5 - eps)[:, None], _lagrange(x, np.clip(2 * x - 1, 0, 1)), 0.0)
        R = np.hstack([left[1:, 1:], right[1:, 1:]])
    else:
        xq, wq = gauss(r + 2)
        V = _lagrange(x, xq)
        M = (V.T * wq) @ V
        Pl, Pr = _lagrange(x, x / 2), _lagrange(x, (x + 1) / 2)
        Minv = np.linalg.inv(M)
        R = np.hstack([Minv @ Pl.T @ M / 2, Minv @ Pr.T @ M / 2])
    n = R.shape[0]
    out = np.zeros((n * nsteps // 2, n * nsteps))
    for it in range(nsteps // 2):
        out[n * it:n * (it + 1), 2 * n * it:2 * n * (it + 1)] = R
    return out


def time_projection(ttype, r_src, r_dst, nsteps=1):
    """get_time_projection_matrix (fe_time.h:749-803): L2 projection between the temporal spaces of one step"""
    xs, xd = _time_nodes(ttype, r_src), _time_nodes(ttype, r_dst)
    xq, wq = gauss(max(r_src, r_dst) + 2)
    Vs, Vd = _lagrange(xs, xq), _lagrange(xd, xq)
    proj = np.linalg.inv((Vd.T * wq) @ Vd) @ ((Vd.T * wq) @ Vs)
    nd, ns = (r_dst + 1, r_src + 1) if ttype == DG else (r_dst, r_src)
    if ttype == DG:
        out = np.zeros((nsteps * nd, nsteps * ns))
        for it in range(nsteps):
            out[it * nd:(it + 1) * nd, it * ns:(it + 1) * ns] = proj
        return out
    full = np.zeros((nsteps * nd + 1, nsteps * ns + 1))  # consecutive steps share their end point (later fills overwrite)
    for it in range(nsteps):
        full[it * nd:it * nd + nd + 1, it * ns:it * ns + ns + 1] = proj
    return full[1:, 1:]
