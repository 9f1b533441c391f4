/*
 * stfem_cpu_baseline.c -- CPU BASELINE (test infrastructure, NOT product code).
 *
 * What bench.py times beside the GPU kernel as "cpu_baseline" (kind "port"): a CPU restatement of
 * the reference's space-time operator apply IN THE REFERENCE'S STRUCTURE, written the way deal.II's
 * MatrixFree executes it, so that the figure is a fair stand-in for the reference binary (deal.II
 * itself is neither under /root/reference nor installed here):
 *
 *   SystemMatrix::vmult (include/operators.h:536-559):   for every source block i
 *       tmp = K src_i;  dst_j += Alpha(j,i) tmp  (skipping exact zeros, 551)
 *       tmp = M src_i;  dst_j += Beta(j,i)  tmp  (556)
 *   MatrixFreeOperator::vmult (1013-1018): zero dst, cell loop
 *   do_cell_integral_local (1135-1173): evaluate (sum factorisation, interpolation to the Gauss
 *       points + collocation derivative) -> quadrature loop -> integrate
 *
 * with what MatrixFree adds on a Cartesian mesh: cell geometry compressed to one diagonal
 * Jacobian (no per-point metric), SIMD ACROSS CELLS (4 cells per AVX2 batch, 5^3 local DoFs in
 * struct-of-arrays form), the element degree a compile-time constant, all granted cores (OpenMP
 * over cell batches of one of 8 colours, so that no two concurrent batches share a DoF).
 * Restricted to what bench.py needs: Cartesian box, zero Dirichlet mask bits as given, no
 * coefficient.  Checked against the oracle in tests/test_cpu_baseline.py.
 */
#include "stfem_oracle.h"

#include <omp.h>
#include <stdlib.h>
#include <string.h>

#define VL 4
typedef double vd __attribute__((vector_size(VL * sizeof(double))));

typedef struct {
  int p, n1, nc[3], nd[3], dmask;
  double h[3];
  double S[8 * 8], D[8 * 8], wq[8]; /* S[q][a], collocation derivative Dq[q][q'] at the Gauss points */
} stfb_plan;

/* derivative of the Lagrange polynomials through the Gauss points, at the Gauss points */
static void collocation_derivative(int n, const double *x, double *Dq)
{
  for (int q = 0; q < n; ++q)
    for (int a = 0; a < n; ++a) {
      double s = 0.0;
      for (int m = 0; m < n; ++m) {
        if (m == a) continue;
        double t = 1.0 / (x[a] - x[m]);
        for (int l = 0; l < n; ++l)
          if (l != a && l != m) t *= (x[q] - x[l]) / (x[a] - x[l]);
        s += t;
      }
      Dq[q * n + a] = s;
    }
}

/* one sum-factorisation sweep over a batch: out[.., o, ..] = sum_i A[o][i] in[.., i, ..] along dir */
#define SWEEP_BODY(N)                                                                         \
  for (int c2 = 0; c2 < N; ++c2)                                                              \
    for (int c1 = 0; c1 < N; ++c1) {                                                          \
      const int base = dir == 0 ? N * (c1 + N * c2) : (dir == 1 ? c1 + N * N * c2 : c1 + N * c2); \
      const int st = dir == 0 ? 1 : (dir == 1 ? N : N * N);                                   \
      vd x[N];                                                                                \
      for (int i = 0; i < N; ++i) x[i] = in[base + i * st];                                   \
      for (int o = 0; o < N; ++o) {                                                           \
        vd s = A[(T ? o : o * N)] * x[0];                                                     \
        for (int i = 1; i < N; ++i) s += A[T ? i * N + o : o * N + i] * x[i];                 \
        out[base + o * st] = s;                                                               \
      }                                                                                       \
    }

#define DEFINE_CELL(N)                                                                                         \
  static inline __attribute__((always_inline)) void sweep##N(const double *A, const int T, const int dir, const vd *in, vd *out) { SWEEP_BODY(N) }        \
  /* K: u -> sum_d (1/h_d^2) S^T.. D^T W D ..S u (vol scaled); M: S^T W S */                                    \
  static void cell_batch##N(const stfb_plan *pl, int lap, const vd *u, vd *r)                                  \
  {                                                                                                            \
    enum { N3 = N * N * N };                                                                                   \
    vd t1[N3], t2[N3], U[N3], G[N3], acc[N3];                                                                  \
    const double vol = pl->h[0] * pl->h[1] * pl->h[2];                                                         \
    sweep##N(pl->S, 0, 0, u, t1);                                                                              \
    sweep##N(pl->S, 0, 1, t1, t2);                                                                             \
    sweep##N(pl->S, 0, 2, t2, U); /* values at the Gauss points */                                             \
    if (!lap) {                                                                                                \
      for (int qz = 0; qz < N; ++qz)                                                                           \
        for (int qy = 0; qy < N; ++qy)                                                                         \
          for (int qx = 0; qx < N; ++qx) U[qx + N * (qy + N * qz)] *= vol * pl->wq[qx] * pl->wq[qy] * pl->wq[qz]; \
      memcpy(acc, U, sizeof(acc));                                                                             \
    } else {                                                                                                   \
      for (int i = 0; i < N3; ++i) acc[i] = (vd){0, 0, 0, 0};                                                  \
      for (int d = 0; d < 3; ++d) { /* collocation gradient, J^-T, JxW, J^-1, transposed derivative */          \
        sweep##N(pl->D, 0, d, U, G);                                                                           \
        const double s = vol / (pl->h[d] * pl->h[d]);                                                          \
        for (int qz = 0; qz < N; ++qz)                                                                         \
          for (int qy = 0; qy < N; ++qy)                                                                       \
            for (int qx = 0; qx < N; ++qx) G[qx + N * (qy + N * qz)] *= s * pl->wq[qx] * pl->wq[qy] * pl->wq[qz]; \
        sweep##N(pl->D, 1, d, G, t1);                                                                          \
        for (int i = 0; i < N3; ++i) acc[i] += t1[i];                                                          \
      }                                                                                                        \
    }                                                                                                          \
    sweep##N(pl->S, 1, 2, acc, t1);                                                                            \
    sweep##N(pl->S, 1, 1, t1, t2);                                                                             \
    sweep##N(pl->S, 1, 0, t2, r);                                                                              \
  }

DEFINE_CELL(2)
DEFINE_CELL(3)
DEFINE_CELL(4)
DEFINE_CELL(5)

typedef long long vi __attribute__((vector_size(VL * sizeof(long long))));
static inline vd loadu(const double *p)
{
  vd v;
  memcpy(&v, p, sizeof v);
  return v;
}
static inline void storeu(double *p, vd v) { memcpy(p, &v, sizeof v); }
/* 4 x 4 transpose (deal.II: vectorized_load_and_transpose / transpose_and_store) */
static inline void transpose4(vd *a, vd *b, vd *c, vd *d)
{
  const vd t0 = __builtin_shuffle(*a, *b, (vi){0, 4, 2, 6}), t1 = __builtin_shuffle(*a, *b, (vi){1, 5, 3, 7});
  const vd t2 = __builtin_shuffle(*c, *d, (vi){0, 4, 2, 6}), t3 = __builtin_shuffle(*c, *d, (vi){1, 5, 3, 7});
  *a = __builtin_shuffle(t0, t2, (vi){0, 1, 4, 5});
  *b = __builtin_shuffle(t1, t3, (vi){0, 1, 4, 5});
  *c = __builtin_shuffle(t0, t2, (vi){2, 3, 6, 7});
  *d = __builtin_shuffle(t1, t3, (vi){2, 3, 6, 7});
}

static int constrained(const stfb_plan *pl, int gx, int gy, int gz)
{
  return ((pl->dmask & 1) && gx == 0) || ((pl->dmask & 2) && gx == pl->nd[0] - 1) || ((pl->dmask & 4) && gy == 0) ||
         ((pl->dmask & 8) && gy == pl->nd[1] - 1) || ((pl->dmask & 16) && gz == 0) || ((pl->dmask & 32) && gz == pl->nd[2] - 1);
}

/* dst = (lap ? K : M) src: zero dst, then the cell loop in 8 colours (operators.h:1013-1018, 1112-1133) */
static void space_vmult(const stfb_plan *pl, int lap, const double *src, double *dst, int threads)
{
  const int N = pl->n1, P = pl->p, N3 = N * N * N;
  const long nx = pl->nd[0], nxy = (long)pl->nd[0] * pl->nd[1], n = nxy * pl->nd[2];
#pragma omp parallel for schedule(static) num_threads(threads)
  for (long i = 0; i < n; ++i) dst[i] = 0.0;
  for (int col = 0; col < 8; ++col) {
    const int ox = col & 1, oy = (col >> 1) & 1, oz = col >> 2;
    const int mx = (pl->nc[0] - ox + 1) / 2, my = (pl->nc[1] - oy + 1) / 2, mz = (pl->nc[2] - oz + 1) / 2;
    const int bx = (mx + VL - 1) / VL; /* batches of VL same-colour cells along x */
    const long nbatch = (long)bx * my * mz;
#pragma omp parallel for schedule(static) num_threads(threads)
    for (long b = 0; b < nbatch; ++b) {
      const int ib = (int)(b % bx), iy = (int)((b / bx) % my), iz = (int)(b / ((long)bx * my));
      const int cy = 2 * iy + oy, cz = 2 * iz + oz;
      int cx[VL], ok[VL];
      for (int l = 0; l < VL; ++l) {
        const int k = ib * VL + l;
        ok[l] = k < mx;
        cx[l] = 2 * (ok[l] ? k : 0) + ox;
      }
      const int boundary = (cy == 0 || cy == pl->nc[1] - 1 || cz == 0 || cz == pl->nc[2] - 1 || cx[0] == 0 ||
                            cx[VL - 1] >= pl->nc[0] - 2 || !ok[VL - 1]);
      /* local DoFs of the batch in struct-of-arrays form (filled lane by lane through plain arrays: element-wise
         writes into vector variables go through store forwarding and stall) */
      double ul[125][VL] __attribute__((aligned(32))), rl[125][VL] __attribute__((aligned(32)));
      vd u[125], r[125];
      /* read_dof_values: gather, constrained DoFs read as 0 */
      if (!boundary && N == 5) { /* rows of 5 contiguous doubles per lane: vector loads + 4 x 4 transposes */
        const double *s0[VL];
        for (int l = 0; l < VL; ++l) s0[l] = src + P * cx[l] + nx * (long)(P * cy) + nxy * (long)(P * cz);
        for (int k = 0; k < N; ++k)
          for (int j = 0; j < N; ++j) {
            const long o = nx * j + nxy * k;
            vd a = loadu(s0[0] + o), b = loadu(s0[1] + o), c = loadu(s0[2] + o), d = loadu(s0[3] + o);
            transpose4(&a, &b, &c, &d);
            double *q = ul[N * (j + N * k)];
            *(vd *)(q) = a;
            *(vd *)(q + VL) = b;
            *(vd *)(q + 2 * VL) = c;
            *(vd *)(q + 3 * VL) = d;
            for (int l = 0; l < VL; ++l) q[4 * VL + l] = s0[l][o + 4];
          }
      } else if (!boundary) {
        for (int l = 0; l < VL; ++l) {
          const double *s0 = src + P * cx[l] + nx * (long)(P * cy) + nxy * (long)(P * cz);
          for (int k = 0; k < N; ++k)
            for (int j = 0; j < N; ++j)
              for (int i = 0; i < N; ++i) ul[i + N * (j + N * k)][l] = s0[i + nx * j + nxy * k];
        }
      } else {
        for (int k = 0; k < N; ++k)
          for (int j = 0; j < N; ++j)
            for (int i = 0; i < N; ++i)
              for (int l = 0; l < VL; ++l) {
                const int gx = P * cx[l] + i, gy = P * cy + j, gz = P * cz + k;
                ul[i + N * (j + N * k)][l] = (!ok[l] || constrained(pl, gx, gy, gz)) ? 0.0 : src[gx + nx * gy + nxy * gz];
              }
      }
      for (int q = 0; q < N3; ++q) u[q] = *(const vd *)ul[q];
      switch (N) {
        case 2: cell_batch2(pl, lap, u, r); break;
        case 3: cell_batch3(pl, lap, u, r); break;
        case 4: cell_batch4(pl, lap, u, r); break;
        default: cell_batch5(pl, lap, u, r); break;
      }
      for (int q = 0; q < N3; ++q) *(vd *)rl[q] = r[q];
      /* distribute_local_to_global: scatter-add, constrained DoFs skipped */
      if (!boundary && N == 5) {
        double *d0[VL];
        for (int l = 0; l < VL; ++l) d0[l] = dst + P * cx[l] + nx * (long)(P * cy) + nxy * (long)(P * cz);
        for (int k = 0; k < N; ++k)
          for (int j = 0; j < N; ++j) {
            const long o = nx * j + nxy * k;
            const double *q = rl[N * (j + N * k)];
            vd a = *(const vd *)(q), b = *(const vd *)(q + VL), c = *(const vd *)(q + 2 * VL), d = *(const vd *)(q + 3 * VL);
            transpose4(&a, &b, &c, &d);
            storeu(d0[0] + o, loadu(d0[0] + o) + a);
            storeu(d0[1] + o, loadu(d0[1] + o) + b);
            storeu(d0[2] + o, loadu(d0[2] + o) + c);
            storeu(d0[3] + o, loadu(d0[3] + o) + d);
            for (int l = 0; l < VL; ++l) d0[l][o + 4] += q[4 * VL + l];
          }
      } else if (!boundary) {
        for (int l = 0; l < VL; ++l) {
          double *d0 = dst + P * cx[l] + nx * (long)(P * cy) + nxy * (long)(P * cz);
          for (int k = 0; k < N; ++k)
            for (int j = 0; j < N; ++j)
              for (int i = 0; i < N; ++i) d0[i + nx * j + nxy * k] += rl[i + N * (j + N * k)][l];
        }
      } else {
        for (int k = 0; k < N; ++k)
          for (int j = 0; j < N; ++j)
            for (int i = 0; i < N; ++i)
              for (int l = 0; l < VL; ++l) {
                if (!ok[l]) continue;
                const int gx = P * cx[l] + i, gy = P * cy + j, gz = P * cz + k;
                if (constrained(pl, gx, gy, gz)) continue;
                dst[gx + nx * gy + nxy * gz] += rl[i + N * (j + N * k)][l];
              }
      }
    }
  }
}

/* One SystemMatrix::vmult in the reference's structure on a Cartesian box.
 * alpha, beta: nb x nb row-major; src, dst: nb arrays of n_dofs doubles; tmp: n_dofs doubles.
 * Returns 0, or -1 for an unsupported degree. */
int stfb_st_vmult(int p, const int ncell[3], const double lower[3], const double upper[3], int dirichlet_mask,
                  int nb, const double *alpha, const double *beta, const double *const *src, double *const *dst,
                  double *tmp, int threads)
{
  if (p < 1 || p > 4 || nb < 1) return -1;
  stfb_plan pl;
  pl.p = p;
  pl.n1 = p + 1;
  pl.dmask = dirichlet_mask;
  for (int d = 0; d < 3; ++d) {
    pl.nc[d] = ncell[d];
    pl.nd[d] = p * ncell[d] + 1;
    pl.h[d] = (upper[d] - lower[d]) / ncell[d];
  }
  double xq[8], Dn[64];
  stfo_gauss(pl.n1, xq, pl.wq);
  stfo_shape_tables(p, pl.n1, pl.S, Dn);
  collocation_derivative(pl.n1, xq, pl.D);
  if (threads < 1) threads = omp_get_max_threads();
  const long n = (long)pl.nd[0] * pl.nd[1] * pl.nd[2];
  for (int j = 0; j < nb; ++j) {
    double *d = dst[j];
#pragma omp parallel for schedule(static) num_threads(threads)
    for (long i = 0; i < n; ++i) d[i] = 0.0;
  }
  for (int i = 0; i < nb; ++i)
    for (int lap = 1; lap >= 0; --lap) {
      const double *w = lap ? alpha : beta;
      int any = 0;
      for (int j = 0; j < nb; ++j) any |= w[j * nb + i] != 0.0;
      if (!any) continue;
      space_vmult(&pl, lap, src[i], tmp, threads);
      for (int j = 0; j < nb; ++j) {
        const double a = w[j * nb + i];
        if (a == 0.0) continue; /* operators.h:551,556 */
        double *d = dst[j];
#pragma omp parallel for schedule(static) num_threads(threads)
        for (long k = 0; k < n; ++k) d[k] += a * tmp[k];
      }
    }
  return 0;
}
