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
