"""CPU restatement (numpy) of the reference's cell-patch Vanka / additive-Schwarz smoother
(PreconditionVanka, reference include/stmg.h:619-907) for the scalar space-time system
A = Alpha (x) K + Beta (x) M.  TEST INFRASTRUCTURE ONLY: imported by tests/ (and tools/vanka_bench.py's
check), never by the product path.

Follows the reference step by step:
  * K_, M_ : the assembled spatial matrices as MatrixFreeTools::compute_matrix builds them with the
    zero-boundary constraints (tests/tp_01.cc:283-299, operators.h:1021-1033): AffineConstraints::
    distribute_local_to_global drops the rows and columns of constrained DoFs and adds every cell's own
    diagonal entry to the global diagonal, i.e. a constrained row keeps the diagonal of the unconstrained
    assembly and nothing else;
  * valence (stmg.h:841-855): number of cells a DoF belongs to;
  * restrict_to_full_matrices_ (compute_block_matrix.h:50-139): entry (i, j) of the cell block = global
    matrix entry of the cell's DoFs i, j, scaled by valence[row i];
  * block of cell c (stmg.h:806-829): B(k + i n, l + j n) = Beta(i, j) M_c(k, l) + Alpha(i, j) K_c(k, l),
    then B.gauss_jordan() (here numpy.linalg.inv);
  * vmult (stmg.h:832-872): dst = sum over cells of scatter(B_c^-1 gather(src)).
Parity status: unpinned - the reference holds no vector or matrix of the smoother; the restatement is
checked by properties (tests/test_vanka_oracle.py: one-cell mesh = exact inverse, symmetry of the
unweighted variant, agreement of the Kronecker construction of the product code with this dense one)."""
import numpy as np

from . import oracle as _o


class VankaOracle:
    def __init__(self, p, ncell, vertices, dirichlet_mask, Alpha, Beta, coef_lap=None, coef_mass=None):
        self.p, self.nc = p, tuple(ncell)
        self.Alpha, self.Beta = np.asarray(Alpha, float), np.asarray(Beta, float)
        n = p + 1
        nd = [p * c + 1 for c in self.nc]
        N = nd[0] * nd[1] * nd[2]
        self.N, self.nd = N, nd
        free = _o.Oracle(p, self.nc, vertices, 0)  # unconstrained assembly
        if coef_lap is not None:
            free.set_coefficient(1, coef_lap)  # (operators.h:1060-1087: the coefficient replaces the scaling)
        if coef_mass is not None:
            free.set_coefficient(0, coef_mass)
        K, M = free.dense(laplace=1.0), free.dense(mass=1.0)
        # constrained DoFs: rows / columns dropped, the diagonal of the unconstrained assembly stays
        con = np.zeros(nd[::-1], bool)  # [z][y][x]
        m = dirichlet_mask
        if m & 1: con[:, :, 0] = True
        if m & 2: con[:, :, -1] = True
        if m & 4: con[:, 0, :] = True
        if m & 8: con[:, -1, :] = True
        if m & 16: con[0, :, :] = True
        if m & 32: con[-1, :, :] = True
        con = con.ravel()
        for A in (K, M):
            d = A.diagonal().copy()
            A[con, :] = 0.0
            A[:, con] = 0.0
            A[con, con] = d[con]
        self.K, self.M, self.constrained = K, M, con
        # cell DoF lists (x fastest inside the cell) and the valence
        self.cells = []
        val = np.zeros(N)
        for cz in range(self.nc[2]):
            for cy in range(self.nc[1]):
                for cx in range(self.nc[0]):
                    k, j, i = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
                    idx = (p * cx + i) + nd[0] * ((p * cy + j) + nd[1] * (p * cz + k))
                    idx = idx.ravel()
                    self.cells.append(idx)
                    val[idx] += 1.0
        self.valence = val
        nb = self.Alpha.shape[0]
        self.blocks = []
        for idx in self.cells:
            Kc = val[idx, None] * K[np.ix_(idx, idx)]
            Mc = val[idx, None] * M[np.ix_(idx, idx)]
            B = np.kron(self.Beta, Mc) + np.kron(self.Alpha, Kc)
            self.blocks.append(np.linalg.inv(B))
        self.nb = nb

    def vmult(self, src):
        src = np.asarray(src, float)
        dst = np.zeros_like(src)
        for idx, Binv in zip(self.cells, self.blocks):
            loc = Binv @ src[:, idx].ravel()
            dst[:, idx] += loc.reshape(self.nb, -1)
        return dst


class StokesVankaOracle:
    """PreconditionVanka in its block form for the two-variable Stokes system (reference include/stmg.h:626-738, 832-872, created as
    in tests/tp_03stokes.cc:537-540, 714-726: K_mask empty, M_mask(0, 0) only), restated the reference's way: the ASSEMBLED matrices
    of the whole mesh (unit vectors through the Stokes oracle, the method of tests/tp_05dgp_support.cc:140-149), strong velocity
    constraints as AffineConstraints::distribute_local_to_global leaves them (row and column dropped, diagonal of the unconstrained
    assembly kept), valence per variable, restriction to the cell's DoFs (compute_block_matrix.h:50-139), the block
        B((i, k), (j, l)) = valence(k) (Alpha(i, j) K_{iv,jv}(k, l) + [iv = jv = 0] Beta(i, j) M(k, l)),
    numpy.linalg.inv for gauss_jordan, vmult = sum over cells of scatter(B_c^-1 gather(src)).  Unpinned (the reference holds no number
    of the smoother)."""

    def __init__(self, ncell, vertices, dirichlet_mask, viscosity, block_variable, Alpha, Beta, weak_mask=0, penalty1=20.0, penalty2=10.0,
                 dg_pressure=False):
        self.nc = tuple(ncell)
        self.var = list(block_variable)
        self.Alpha, self.Beta = np.asarray(Alpha, float), np.asarray(Beta, float)
        free = _o.StokesOracle(self.nc, vertices, 0, viscosity, weak_mask=weak_mask, penalty1=penalty1, penalty2=penalty2, dg_pressure=dg_pressure)
        nu_, np_ = free.n_u, free.n_p
        self.n_u, self.n_p = nu_, np_
        n = 3 * nu_ + np_
        K = np.zeros((n, n))
        M = np.zeros((3 * nu_, 3 * nu_))
        e = np.zeros(n)
        for j in range(n):
            e[j] = 1.0
            ou, op = free.apply(e[:3 * nu_], e[3 * nu_:], 1.0, 0.0)
            K[:3 * nu_, j], K[3 * nu_:, j] = ou.reshape(-1), op
            if j < 3 * nu_:
                mu, _ = free.apply(e[:3 * nu_], np.zeros(np_), 0.0, 1.0)
                M[:, j] = mu.reshape(-1)
            e[j] = 0.0
        ndu = [2 * c + 1 for c in self.nc]
        con = np.zeros(ndu[::-1], bool)
        m = dirichlet_mask
        if m & 1: con[:, :, 0] = True
        if m & 2: con[:, :, -1] = True
        if m & 4: con[:, 0, :] = True
        if m & 8: con[:, -1, :] = True
        if m & 16: con[0, :, :] = True
        if m & 32: con[-1, :, :] = True
        con = np.concatenate([np.tile(con.ravel(), 3), np.zeros(np_, bool)])
        d = K.diagonal().copy()
        K[con, :] = 0.0
        K[:, con] = 0.0
        K[con, con] = d[con]
        cu = con[:3 * nu_]
        d = M.diagonal().copy()
        M[cu, :] = 0.0
        M[:, cu] = 0.0
        M[cu, cu] = d[cu]
        # cell DoF lists per variable and the valences
        ndp = [c + 1 for c in self.nc]
        self.cells = []
        valu, valp = np.zeros(3 * nu_), np.zeros(np_)
        cell = 0
        for cz in range(self.nc[2]):
            for cy in range(self.nc[1]):
                for cx in range(self.nc[0]):
                    k, j, i = np.meshgrid(np.arange(3), np.arange(3), np.arange(3), indexing="ij")
                    iu = ((2 * cx + i) + ndu[0] * ((2 * cy + j) + ndu[1] * (2 * cz + k))).ravel()
                    iu = np.concatenate([c * nu_ + iu for c in range(3)])
                    if dg_pressure:
                        ip = 4 * cell + np.arange(4)
                    else:
                        k, j, i = np.meshgrid(np.arange(2), np.arange(2), np.arange(2), indexing="ij")
                        ip = ((cx + i) + ndp[0] * ((cy + j) + ndp[1] * (cz + k))).ravel()
                    self.cells.append((iu, ip))
                    valu[iu] += 1.0
                    valp[ip] += 1.0
                    cell += 1
        nblk = len(self.var)
        self.blocks = []
        for iu, ip in self.cells:
            idx = [iu, 3 * nu_ + ip]          # rows of K per variable
            val = [valu[iu], valp[ip]]
            size = [len(iu), len(ip)]
            off = np.concatenate([[0], np.cumsum([size[v] for v in self.var])])
            B = np.zeros((off[-1], off[-1]))
            for bi in range(nblk):
                for bj in range(nblk):
                    iv, jv = self.var[bi], self.var[bj]
                    blk = self.Alpha[bi, bj] * K[np.ix_(idx[iv], idx[jv])]
                    if iv == 0 and jv == 0:
                        blk = blk + self.Beta[bi, bj] * M[np.ix_(iu, iu)]
                    B[off[bi]:off[bi + 1], off[bj]:off[bj + 1]] = val[iv][:, None] * blk
            self.blocks.append(np.linalg.inv(B))

    def vmult(self, src_blocks):
        """src_blocks: list of arrays in BlockSlice order (velocity 3 n_u, pressure n_p); returns the same shapes"""
        src = [np.asarray(b, float).reshape(-1) for b in src_blocks]
        dst = [np.zeros_like(b) for b in src]
        for (iu, ip), Binv in zip(self.cells, self.blocks):
            loc = np.concatenate([src[b][iu if v == 0 else ip] for b, v in enumerate(self.var)])
            y = Binv @ loc
            o = 0
            for b, v in enumerate(self.var):
                ii = iu if v == 0 else ip
                dst[b][ii] += y[o:o + len(ii)]
                o += len(ii)
        return dst
