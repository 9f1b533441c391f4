/*
 * stfem_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See stfem_oracle.h for scope, parity status and the reference file:line map.
 */
#include "stfem_oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXN 12 /* max 1D points (p+1) */

/* ------------------------------------------------------------------------- */
/* 1D quadrature rules (deal.II QGauss / QGaussLobatto / QGaussRadau on [0,1]) */
/* ------------------------------------------------------------------------- */

static void legendre(int n, double x, double *pn, double *pnm1)
{
  double p0 = 1.0, p1 = x;
  if (n == 0) { *pn = 1.0; *pnm1 = 0.0; return; }
  for (int k = 2; k <= n; ++k) {
    double pk = ((2.0 * k - 1.0) * x * p1 - (k - 1.0) * p0) / k;
    p0 = p1; p1 = pk;
  }
  *pn = p1; *pnm1 = p0;
}

static double legendre_deriv(int n, double x)
{
  double pn, pnm1;
  legendre(n, x, &pn, &pnm1);
  return n * (x * pn - pnm1) / (x * x - 1.0);
}

/* kind 0: P_n ; kind 1: P_n' ; kind 2: (P_n - P_{n-1}) */
static double rootfun(int kind, int n, double x)
{
  double pn, pnm1;
  if (kind == 1) return legendre_deriv(n, x);
  legendre(n, x, &pn, &pnm1);
  return kind == 0 ? pn : pn - pnm1;
}

/* all simple roots of rootfun in the open interval (-1,1), ascending */
static int find_roots(int kind, int n, double *roots)
{
  const int ngrid = 20000;
  int nr = 0;
  const double lo = -1.0 + 1e-9, hi = 1.0 - 1e-9;
  double xa = lo, fa = rootfun(kind, n, xa);
  for (int i = 1; i <= ngrid; ++i) {
    double xb = lo + (hi - lo) * i / ngrid, fb = rootfun(kind, n, xb);
    if (fa == 0.0) { roots[nr++] = xa; }
    else if (fa * fb < 0.0) {
      double a = xa, b = xb, ga = fa;
      for (int it = 0; it < 200; ++it) {
        double m = 0.5 * (a + b), gm = rootfun(kind, n, m);
        if (gm == 0.0) { a = b = m; break; }
        if (ga * gm < 0.0) b = m; else { a = m; ga = gm; }
        if (b - a < 1e-17) break;
      }
      roots[nr++] = 0.5 * (a + b);
    }
    xa = xb; fa = fb;
  }
  return nr;
}

void stfo_gauss(int n, double *x, double *w)
{
  double r[MAXN + 4];
  int nr = find_roots(0, n, r);
  (void)nr;
  for (int i = 0; i < n; ++i) {
    /* enforce symmetry */
    double xi = 0.5 * (r[i] - r[n - 1 - i]);
    double d = legendre_deriv(n, xi);
    x[i] = 0.5 * (xi + 1.0);
    w[i] = 1.0 / ((1.0 - xi * xi) * d * d);
  }
}

void stfo_gauss_lobatto(int n, double *x)
{
  double r[MAXN + 4];
  x[0] = 0.0; x[n - 1] = 1.0;
  if (n > 2) {
    find_roots(1, n - 1, r);
    for (int i = 0; i < n - 2; ++i) {
      double xi = 0.5 * (r[i] - r[n - 3 - i]);
      x[i + 1] = 0.5 * (xi + 1.0);
    }
  }
}

void stfo_gauss_radau_right(int n, double *x)
{
  double r[MAXN + 4];
  if (n > 1) find_roots(2, n, r); /* roots of P_n - P_{n-1} other than +1 */
  for (int i = 0; i < n - 1; ++i) x[i] = 0.5 * (r[i] + 1.0);
  x[n - 1] = 1.0;
}

static double lagrange(int n, const double *xi, int a, double x)
{
  double v = 1.0;
  for (int b = 0; b < n; ++b)
    if (b != a) v *= (x - xi[b]) / (xi[a] - xi[b]);
  return v;
}

static double lagrange_deriv(int n, const double *xi, int a, double x)
{
  double s = 0.0;
  for (int c = 0; c < n; ++c) {
    if (c == a) continue;
    double v = 1.0 / (xi[a] - xi[c]);
    for (int b = 0; b < n; ++b)
      if (b != a && b != c) v *= (x - xi[b]) / (xi[a] - xi[b]);
    s += v;
  }
  return s;
}

void stfo_shape_tables(int p, int nq, double *S, double *D)
{
  double xi[MAXN], xq[MAXN], wq[MAXN];
  const int n1 = p + 1;
  stfo_gauss_lobatto(n1, xi);
  stfo_gauss(nq, xq, wq);
  for (int q = 0; q < nq; ++q)
    for (int a = 0; a < n1; ++a) {
      S[q * n1 + a] = lagrange(n1, xi, a, xq[q]);
      D[q * n1 + a] = lagrange_deriv(n1, xi, a, xq[q]);
    }
}

/* the same tables at arbitrary points of [0, 1] (face evaluation: the end points 0 and 1) */
void stfo_shape_tables_at(int p, int npts, const double *pts, double *S, double *D)
{
  double xi[MAXN];
  const int n1 = p + 1;
  stfo_gauss_lobatto(n1, xi);
  for (int q = 0; q < npts; ++q)
    for (int a = 0; a < n1; ++a) {
      S[q * n1 + a] = lagrange(n1, xi, a, pts[q]);
      D[q * n1 + a] = lagrange_deriv(n1, xi, a, pts[q]);
    }
}

/* ------------------------------------------------------------------------- */
/* spatial operator                                                           */
/* ------------------------------------------------------------------------- */

struct stfo_ctx {
  int p, n1, nq;
  int nc[3];
  int nd[3];
  long ndofs, ncells;
  int dirichlet_mask;
  double *vertices;
  double S[MAXN * MAXN], D[MAXN * MAXN];   /* [q][a] */
  double St[MAXN * MAXN], Dt[MAXN * MAXN]; /* [a][q] */
  double xq[MAXN], wq[MAXN];
  unsigned char *constrained;
  double *coef_mass, *coef_lap;
};

static int g_threads = 0;
void stfo_set_threads(int n) { g_threads = n; }

stfo_ctx *stfo_create(int p, const int ncell[3], const double *vertices, int dirichlet_mask)
{
  if (p < 1 || p + 1 > MAXN) return NULL;
  stfo_ctx *c = (stfo_ctx *)calloc(1, sizeof(stfo_ctx));
  c->p = p; c->n1 = p + 1; c->nq = p + 1;
  c->ncells = 1; c->ndofs = 1;
  for (int d = 0; d < 3; ++d) {
    c->nc[d] = ncell[d];
    c->nd[d] = p * ncell[d] + 1;
    c->ncells *= ncell[d];
    c->ndofs *= c->nd[d];
  }
  c->dirichlet_mask = dirichlet_mask;
  long nv = (long)(ncell[0] + 1) * (ncell[1] + 1) * (ncell[2] + 1);
  c->vertices = (double *)malloc(sizeof(double) * 3 * nv);
  memcpy(c->vertices, vertices, sizeof(double) * 3 * nv);
  stfo_shape_tables(p, c->nq, c->S, c->D);
  for (int q = 0; q < c->nq; ++q)
    for (int a = 0; a < c->n1; ++a) {
      c->St[a * c->nq + q] = c->S[q * c->n1 + a];
      c->Dt[a * c->nq + q] = c->D[q * c->n1 + a];
    }
  stfo_gauss(c->nq, c->xq, c->wq);
  c->constrained = (unsigned char *)calloc(c->ndofs, 1);
  for (int k = 0; k < c->nd[2]; ++k)
    for (int j = 0; j < c->nd[1]; ++j)
      for (int i = 0; i < c->nd[0]; ++i) {
        int f = 0;
        if ((dirichlet_mask & 1) && i == 0) f = 1;
        if ((dirichlet_mask & 2) && i == c->nd[0] - 1) f = 1;
        if ((dirichlet_mask & 4) && j == 0) f = 1;
        if ((dirichlet_mask & 8) && j == c->nd[1] - 1) f = 1;
        if ((dirichlet_mask & 16) && k == 0) f = 1;
        if ((dirichlet_mask & 32) && k == c->nd[2] - 1) f = 1;
        c->constrained[i + (long)c->nd[0] * (j + (long)c->nd[1] * k)] = (unsigned char)f;
      }
  return c;
}

void stfo_destroy(stfo_ctx *c)
{
  if (!c) return;
  free(c->vertices); free(c->constrained); free(c->coef_mass); free(c->coef_lap);
  free(c);
}

long stfo_n_dofs(const stfo_ctx *c) { return c->ndofs; }
long stfo_n_cells(const stfo_ctx *c) { return c->ncells; }
int stfo_n_q(const stfo_ctx *c) { return c->nq; }

void stfo_set_coefficient(stfo_ctx *c, int which, const double *coef)
{
  double **slot = which == 0 ? &c->coef_mass : &c->coef_lap;
  free(*slot); *slot = NULL;
  if (coef) {
    size_t n = (size_t)c->ncells * c->nq * c->nq * c->nq;
    *slot = (double *)malloc(n * sizeof(double));
    memcpy(*slot, coef, n * sizeof(double));
  }
}

static void cell_vertices(const stfo_ctx *c, int cx, int cy, int cz, double v[8][3])
{
  const long nvx = c->nc[0] + 1, nvy = c->nc[1] + 1;
  for (int k = 0; k < 2; ++k)
    for (int j = 0; j < 2; ++j)
      for (int i = 0; i < 2; ++i) {
        long idx = (cx + i) + nvx * ((cy + j) + nvy * (long)(cz + k));
        for (int d = 0; d < 3; ++d) v[i + 2 * j + 4 * k][d] = c->vertices[3 * idx + d];
      }
}

/* MappingQ1: x(xi) trilinear; J[d][e] = d x_d / d xi_e */
static void trilinear(const double v[8][3], double x, double y, double z, double pt[3],
                      double J[3][3])
{
  const double fx[2] = {1.0 - x, x}, fy[2] = {1.0 - y, y}, fz[2] = {1.0 - z, z};
  const double dx[2] = {-1.0, 1.0};
  for (int d = 0; d < 3; ++d) { pt[d] = 0; J[d][0] = J[d][1] = J[d][2] = 0; }
  for (int k = 0; k < 2; ++k)
    for (int j = 0; j < 2; ++j)
      for (int i = 0; i < 2; ++i) {
        const double *X = v[i + 2 * j + 4 * k];
        for (int d = 0; d < 3; ++d) {
          pt[d] += X[d] * fx[i] * fy[j] * fz[k];
          J[d][0] += X[d] * dx[i] * fy[j] * fz[k];
          J[d][1] += X[d] * fx[i] * dx[j] * fz[k];
          J[d][2] += X[d] * fx[i] * fy[j] * dx[k];
        }
      }
}

static double invert3(const double J[3][3], double Ji[3][3])
{
  const double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) -
                     J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
                     J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
  const double id = 1.0 / det;
  Ji[0][0] = (J[1][1] * J[2][2] - J[1][2] * J[2][1]) * id;
  Ji[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id;
  Ji[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
  Ji[1][0] = (J[1][2] * J[2][0] - J[1][0] * J[2][2]) * id;
  Ji[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id;
  Ji[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
  Ji[2][0] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) * id;
  Ji[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id;
  Ji[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
  return det;
}

void stfo_quadrature_points(const stfo_ctx *c, double *out)
{
  const int nq = c->nq;
  for (int cz = 0; cz < c->nc[2]; ++cz)
    for (int cy = 0; cy < c->nc[1]; ++cy)
      for (int cx = 0; cx < c->nc[0]; ++cx) {
        long cell = cx + (long)c->nc[0] * (cy + (long)c->nc[1] * cz);
        double v[8][3], J[3][3];
        cell_vertices(c, cx, cy, cz, v);
        for (int qz = 0; qz < nq; ++qz)
          for (int qy = 0; qy < nq; ++qy)
            for (int qx = 0; qx < nq; ++qx) {
              long q = qx + nq * (qy + nq * qz);
              trilinear(v, c->xq[qx], c->xq[qy], c->xq[qz], out + 3 * (cell * nq * nq * nq + q), J);
            }
      }
}

/* out(.., o, ..) = sum_i A[o*nin+i] in(.., i, ..) along direction dir; x fastest */
static void sweep(const double *A, int nout, int nin, int dir, const int din[3], const double *in,
                  double *out)
{
  int dout[3] = {din[0], din[1], din[2]};
  dout[dir] = nout;
  long sin[3] = {1, din[0], (long)din[0] * din[1]};
  long sout[3] = {1, dout[0], (long)dout[0] * dout[1]};
  int e1 = (dir + 1) % 3, e2 = (dir + 2) % 3;
  for (int a = 0; a < din[e2]; ++a)
    for (int b = 0; b < din[e1]; ++b) {
      const double *pi = in + a * sin[e2] + b * sin[e1];
      double *po = out + a * sout[e2] + b * sout[e1];
      for (int o = 0; o < nout; ++o) {
        double s = 0.0;
        for (int i = 0; i < nin; ++i) s += A[o * nin + i] * pi[i * sin[dir]];
        po[o * sout[dir]] = s;
      }
    }
}

/* operators.h:1135-1173: evaluate -> quadrature loop -> integrate, one cell */
static void cell_apply(const stfo_ctx *c, int cx, int cy, int cz, double ms, double ls,
                       const double *u, double *r)
{
  const int n1 = c->n1, nq = c->nq;
  const int NQ3 = nq * nq * nq;
  const long cell = cx + (long)c->nc[0] * (cy + (long)c->nc[1] * cz);
  double t1[MAXN * MAXN * MAXN], t2[MAXN * MAXN * MAXN], t3[MAXN * MAXN * MAXN];
  double U[MAXN * MAXN * MAXN], G[3][MAXN * MAXN * MAXN];
  const int do_mass = (ms != 0.0), do_lap = (ls != 0.0);
  int d0[3] = {n1, n1, n1}, d1[3] = {nq, n1, n1}, d2[3] = {nq, nq, n1};

  /* evaluate: values and reference gradients at the Gauss points */
  sweep(c->S, nq, n1, 0, d0, u, t1);  /* Sx u */
  sweep(c->S, nq, n1, 1, d1, t1, t2); /* Sy Sx u */
  if (do_mass) sweep(c->S, nq, n1, 2, d2, t2, U);
  if (do_lap) {
    sweep(c->D, nq, n1, 2, d2, t2, G[2]); /* Dz Sy Sx u */
    sweep(c->D, nq, n1, 1, d1, t1, t3);
    sweep(c->S, nq, n1, 2, d2, t3, G[1]); /* Sz Dy Sx u */
    sweep(c->D, nq, n1, 0, d0, u, t1);
    sweep(c->S, nq, n1, 1, d1, t1, t3);
    sweep(c->S, nq, n1, 2, d2, t3, G[0]); /* Sz Sy Dx u */
  }

  /* quadrature loop (operators.h:1149-1163); coefficient REPLACES the scaling */
  double v[8][3];
  cell_vertices(c, cx, cy, cz, v);
  for (int qz = 0; qz < nq; ++qz)
    for (int qy = 0; qy < nq; ++qy)
      for (int qx = 0; qx < nq; ++qx) {
        const int q = qx + nq * (qy + nq * qz);
        double pt[3], J[3][3], Ji[3][3];
        trilinear(v, c->xq[qx], c->xq[qy], c->xq[qz], pt, J);
        const double det = invert3(J, Ji);
        const double JxW = det * c->wq[qx] * c->wq[qy] * c->wq[qz];
        if (do_mass) {
          const double cm = c->coef_mass ? c->coef_mass[cell * NQ3 + q] : ms;
          U[q] = cm * U[q] * JxW;
        }
        if (do_lap) {
          const double cl = c->coef_lap ? c->coef_lap[cell * NQ3 + q] : ls;
          double gp[3], gr[3] = {G[0][q], G[1][q], G[2][q]};
          for (int i = 0; i < 3; ++i) /* J^{-T} grad_ref */
            gp[i] = Ji[0][i] * gr[0] + Ji[1][i] * gr[1] + Ji[2][i] * gr[2];
          for (int i = 0; i < 3; ++i) gp[i] *= cl * JxW;
          for (int e = 0; e < 3; ++e) /* J^{-1} flux */
            G[e][q] = Ji[e][0] * gp[0] + Ji[e][1] * gp[1] + Ji[e][2] * gp[2];
        }
      }

  /* integrate: transposed sweeps */
  int e0[3] = {nq, nq, nq}, e1[3] = {nq, nq, n1}, e2[3] = {nq, n1, n1};
  const int N3 = n1 * n1 * n1;
  for (int i = 0; i < N3; ++i) r[i] = 0.0;
  if (do_mass) {
    sweep(c->St, n1, nq, 2, e0, U, t1);
    sweep(c->St, n1, nq, 1, e1, t1, t2);
    sweep(c->St, n1, nq, 0, e2, t2, t3);
    for (int i = 0; i < N3; ++i) r[i] += t3[i];
  }
  if (do_lap) {
    sweep(c->Dt, n1, nq, 2, e0, G[2], t1);
    sweep(c->St, n1, nq, 1, e1, t1, t2);
    sweep(c->St, n1, nq, 0, e2, t2, t3);
    for (int i = 0; i < N3; ++i) r[i] += t3[i];
    sweep(c->St, n1, nq, 2, e0, G[1], t1);
    sweep(c->Dt, n1, nq, 1, e1, t1, t2);
    sweep(c->St, n1, nq, 0, e2, t2, t3);
    for (int i = 0; i < N3; ++i) r[i] += t3[i];
    sweep(c->St, n1, nq, 2, e0, G[0], t1);
    sweep(c->St, n1, nq, 1, e1, t1, t2);
    sweep(c->Dt, n1, nq, 0, e2, t2, t3);
    for (int i = 0; i < N3; ++i) r[i] += t3[i];
  }
}

static inline long dof_index(const stfo_ctx *c, int cx, int cy, int cz, int a, int b, int g)
{
  return (c->p * cx + a) + (long)c->nd[0] * ((c->p * cy + b) + (long)c->nd[1] * (c->p * cz + g));
}

/* operators.h:1013-1018 + 1112-1133: dst zeroed, gather (constrained -> 0),
 * cell integral, scatter-add (constrained rows skipped).  Cells of one 2x2x2
 * parity colour share no DoF, so a colour is an OpenMP-parallel loop. */
void stfo_space_vmult(const stfo_ctx *c, double ms, double ls, double *dst, const double *src)
{
  const int n1 = c->n1;
  memset(dst, 0, sizeof(double) * c->ndofs);
#ifdef _OPENMP
  const int nthreads = g_threads > 0 ? g_threads : omp_get_max_threads();
#endif
  for (int colour = 0; colour < 8; ++colour) {
    const int ox = colour & 1, oy = (colour >> 1) & 1, oz = (colour >> 2) & 1;
    const int mx = (c->nc[0] - ox + 1) / 2, my = (c->nc[1] - oy + 1) / 2,
              mz = (c->nc[2] - oz + 1) / 2;
    const long ncol = (long)mx * my * mz;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads)
#endif
    for (long t = 0; t < ncol; ++t) {
      const int cx = 2 * (int)(t % mx) + ox, cy = 2 * (int)((t / mx) % my) + oy,
                cz = 2 * (int)(t / ((long)mx * my)) + oz;
      double u[MAXN * MAXN * MAXN], r[MAXN * MAXN * MAXN];
      for (int g = 0; g < n1; ++g)
        for (int b = 0; b < n1; ++b)
          for (int a = 0; a < n1; ++a) {
            long gi = dof_index(c, cx, cy, cz, a, b, g);
            u[a + n1 * (b + n1 * g)] = c->constrained[gi] ? 0.0 : src[gi];
          }
      cell_apply(c, cx, cy, cz, ms, ls, u, r);
      for (int g = 0; g < n1; ++g)
        for (int b = 0; b < n1; ++b)
          for (int a = 0; a < n1; ++a) {
            long gi = dof_index(c, cx, cy, cz, a, b, g);
            if (!c->constrained[gi]) dst[gi] += r[a + n1 * (b + n1 * g)];
          }
    }
  }
}

/* operators.h:1092-1110 (forward diagonal only; constrained rows stay 0,
 * consistent with vmult never writing them) */
void stfo_diagonal(const stfo_ctx *c, double ms, double ls, double *diag)
{
  const int n1 = c->n1, N3 = n1 * n1 * n1;
  memset(diag, 0, sizeof(double) * c->ndofs);
  for (int cz = 0; cz < c->nc[2]; ++cz)
    for (int cy = 0; cy < c->nc[1]; ++cy)
      for (int cx = 0; cx < c->nc[0]; ++cx) {
        double u[MAXN * MAXN * MAXN], r[MAXN * MAXN * MAXN];
        for (int l = 0; l < N3; ++l) {
          memset(u, 0, sizeof(double) * N3);
          u[l] = 1.0;
          cell_apply(c, cx, cy, cz, ms, ls, u, r);
          long gi = dof_index(c, cx, cy, cz, l % n1, (l / n1) % n1, l / (n1 * n1));
          if (!c->constrained[gi]) diag[gi] += r[l];
        }
      }
}

void stfo_dense(const stfo_ctx *c, double ms, double ls, double *A)
{
  const long n = c->ndofs;
  double *e = (double *)calloc(n, sizeof(double));
  double *y = (double *)malloc(n * sizeof(double));
  for (long j = 0; j < n; ++j) {
    e[j] = 1.0;
    stfo_space_vmult(c, ms, ls, y, e);
    e[j] = 0.0;
    for (long i = 0; i < n; ++i) A[i * n + j] = y[i];
  }
  free(e); free(y);
}

/* operators.h:536-611.  K = MatrixFreeOperator(.,0,1), M = (.,1,0)
 * (tests/tp_01.cc:114-117). */
void stfo_st_vmult(const stfo_ctx *c, int nrows, int ncols, const double *alpha,
                   const double *beta, int transpose, int add, double *const *dst,
                   const double *const *src)
{
  const long n = c->ndofs;
  const int nsrc = transpose ? nrows : ncols, ndst = transpose ? ncols : nrows;
  double *tmp = (double *)malloc(n * sizeof(double));
  if (!add)
    for (int j = 0; j < ndst; ++j) memset(dst[j], 0, n * sizeof(double));
  for (int i = 0; i < nsrc; ++i) {
    stfo_space_vmult(c, 0.0, 1.0, tmp, src[i]);
    for (int j = 0; j < ndst; ++j) {
      const double a = transpose ? alpha[i * ncols + j] : alpha[j * ncols + i];
      if (a != 0.0)
        for (long k = 0; k < n; ++k) dst[j][k] += a * tmp[k];
    }
    stfo_space_vmult(c, 1.0, 0.0, tmp, src[i]);
    for (int j = 0; j < ndst; ++j) {
      const double b = transpose ? beta[i * ncols + j] : beta[j * ncols + i];
      if (b != 0.0)
        for (long k = 0; k < n; ++k) dst[j][k] += b * tmp[k];
    }
  }
  free(tmp);
}

/* ------------------------------------------------------------------------- */
/* Coefficient (operators.h:870-965)                                          */
/* ------------------------------------------------------------------------- */

/* std/boost mt19937 (32 bit), default seed 5489 */
typedef struct { uint32_t mt[624]; int idx; } mt19937_t;
static void mt_seed(mt19937_t *s, uint32_t seed)
{
  s->mt[0] = seed;
  for (int i = 1; i < 624; ++i)
    s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
  s->idx = 624;
}
static uint32_t mt_next(mt19937_t *s)
{
  if (s->idx >= 624) {
    for (int i = 0; i < 624; ++i) {
      uint32_t y = (s->mt[i] & 0x80000000u) | (s->mt[(i + 1) % 624] & 0x7fffffffu);
      s->mt[i] = s->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    s->idx = 0;
  }
  uint32_t y = s->mt[s->idx++];
  y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
  return y;
}

/* boost::random::uniform_real_distribution<double> on a 32-bit integer engine
 * (published algorithm of boost/random/uniform_real_distribution.hpp: one draw,
 * numerator/(range+1)*(max-min)+min, redraw if == max).  Boost is absent here,
 * so this stream is restated, not verified against a boost build. */
static double boost_uniform_real(mt19937_t *s, double lo, double hi)
{
  for (;;) {
    double num = (double)mt_next(s);
    double res = num / 4294967296.0 * (hi - lo) + lo;
    if (res < hi) return res;
  }
}

void stfo_coefficient_values(const stfo_ctx *c, double c1, double c2, double c3, double distort,
                             const int sub[3], const double lower[3], const double upper[3],
                             double *out)
{
  const int nq = c->nq;
  const long NQ3 = (long)nq * nq * nq;
  double *pts = (double *)malloc(sizeof(double) * 3 * c->ncells * NQ3);
  double *table = NULL, step[3] = {0, 0, 0};
  stfo_quadrature_points(c, pts);
  if (distort != 0.0) {
    long nt = (long)sub[0] * sub[1] * sub[2];
    table = (double *)malloc(sizeof(double) * nt);
    mt19937_t rng;
    mt_seed(&rng, 5489u);
    for (long i = 0; i < nt; ++i) table[i] = boost_uniform_real(&rng, 1 - distort, 1 + distort);
    for (int d = 0; d < 3; ++d) step[d] = (upper[d] - lower[d]) / sub[d];
  }
  for (long i = 0; i < c->ncells * NQ3; ++i) {
    const double px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
    double v = c1;
    if (py >= 0.2) v = (px < 0.2) ? c2 : c3;
    if (table) {
      unsigned ix = (unsigned)((px - lower[0]) / step[0]);
      unsigned iy = (unsigned)((py - lower[1]) / step[1]);
      unsigned iz = (unsigned)((pz - lower[2]) / step[2]);
      /* Table<3>::fill is C-style: last index fastest */
      v *= table[((long)ix * sub[1] + iy) * sub[2] + iz];
    }
    out[i] = v;
  }
  free(pts); free(table);
}

/* ------------------------------------------------------------------------- */
/* temporal matrices (fe_time.h)                                              */
/* ------------------------------------------------------------------------- */

int stfo_time_nb(int type, int r, int nsteps) { return (type == 0 ? r : r + 1) * nsteps; }

/* fe_time.h:643-696 */
int stfo_cg_weights(int r, double *M, double *Dm)
{
  double xt[MAXN], xq[MAXN + 2], wq[MAXN + 2];
  const int nt = r + 1, nq = r + 2;
  if (r < 1 || nq > MAXN) return -1;
  stfo_gauss_lobatto(nt, xt);
  stfo_gauss(nq, xq, wq);
  const double *xtest = xt + 1; /* trial points minus the first */
  for (int i = 0; i < r; ++i)
    for (int j = 0; j < nt; ++j) {
      double m = 0, d = 0;
      for (int q = 0; q < nq; ++q) {
        const double psi = lagrange(r, xtest, i, xq[q]);
        m += wq[q] * psi * lagrange(nt, xt, j, xq[q]);
        d += wq[q] * psi * lagrange_deriv(nt, xt, j, xq[q]);
      }
      M[i * nt + j] = m;
      Dm[i * nt + j] = d;
    }
  return 0;
}

/* fe_time.h:698-744 */
int stfo_dg_weights(int r, double *M, double *Dm, double *jump)
{
  double xt[MAXN], xq[MAXN + 2], wq[MAXN + 2];
  const int nt = r + 1, nq = r + 2;
  if (r < 0 || nq > MAXN) return -1;
  stfo_gauss_radau_right(nt, xt);
  stfo_gauss(nq, xq, wq);
  for (int i = 0; i < nt; ++i) {
    jump[i] = lagrange(nt, xt, i, 0.0);
    for (int j = 0; j < nt; ++j) {
      double m = 0, d = lagrange(nt, xt, i, 0.0) * lagrange(nt, xt, j, 0.0);
      for (int q = 0; q < nq; ++q) {
        const double phi = lagrange(nt, xt, i, xq[q]);
        m += wq[q] * phi * lagrange(nt, xt, j, xq[q]);
        d += wq[q] * phi * lagrange_deriv(nt, xt, j, xq[q]);
      }
      M[i * nt + j] = m;
      Dm[i * nt + j] = d;
    }
  }
  return 0;
}

/* single-step Alpha,Beta,Gamma,Zeta exactly as get_fe_time_weights returns
 * them for n_timesteps_at_once=1 (fe_time.h:351-409 incl. the DG swap) */
static int weights_1(int type, int r, double tau, double *A, double *B, double *G, double *Z,
                     double *t2, double *t3)
{
  double M[MAXN * MAXN], Dm[MAXN * MAXN], jump[MAXN];
  const int nt = type == 0 ? r : r + 1;
  if (type == 0) {
    if (stfo_cg_weights(r, M, Dm)) return -1;
    /* split_lhs_rhs, fe_time.h:485-504 */
    for (int i = 0; i < nt; ++i) {
      for (int j = 0; j < nt; ++j) {
        A[i * nt + j] = tau * M[i * (r + 1) + j + 1];
        B[i * nt + j] = Dm[i * (r + 1) + j + 1];
      }
      t2[i] = -tau * M[i * (r + 1)];
      t3[i] = -Dm[i * (r + 1)];
      G[i] = t2[i];
      Z[i] = t3[i];
    }
  } else {
    if (stfo_dg_weights(r, M, Dm, jump)) return -1;
    for (int i = 0; i < nt; ++i) {
      for (int j = 0; j < nt; ++j) {
        A[i * nt + j] = tau * M[i * nt + j];
        B[i * nt + j] = Dm[i * nt + j];
      }
      t2[i] = 0.0;     /* tmp[2] = 0 */
      t3[i] = jump[i]; /* tmp[3] = jump */
      G[i] = t3[i];    /* ret[2] = tmp[3] for DG */
      Z[i] = t2[i];    /* ret[3] = tmp[2] for DG */
    }
  }
  return nt;
}

int stfo_time_weights(int type, int r, double tau, int nsteps, double *Alpha, double *Beta,
                      double *Gamma, double *Zeta)
{
  double A[MAXN * MAXN], B[MAXN * MAXN], G[MAXN], Z[MAXN], t2[MAXN], t3[MAXN];
  const int nt = weights_1(type, r, tau, A, B, G, Z, t2, t3);
  if (nt < 0) return -1;
  const int nb = nt * nsteps;
  memset(Alpha, 0, sizeof(double) * nb * nb);
  memset(Beta, 0, sizeof(double) * nb * nb);
  memset(Gamma, 0, sizeof(double) * nb);
  memset(Zeta, 0, sizeof(double) * nb);
  for (int it = 0; it < nsteps; ++it)
    for (int i = 0; i < nt; ++i) {
      if (it < nsteps - 1 && i == nt - 1)
        for (int j = 0; j < nt; ++j) {
          Alpha[(j + (it + 1) * nt) * nb + i + it * nt] = -t2[j];
          Beta[(j + (it + 1) * nt) * nb + i + it * nt] = -t3[j];
        }
      for (int j = 0; j < nt; ++j) {
        Alpha[(i + it * nt) * nb + j + it * nt] = A[i * nt + j];
        Beta[(i + it * nt) * nb + j + it * nt] = B[i * nt + j];
      }
    }
  for (int i = 0; i < nt; ++i) { Gamma[i] = G[i]; Zeta[i] = Z[i]; }
  return nb;
}

static void mat_inv(int n, const double *A, double *Ai)
{
  double a[MAXN * MAXN * 2];
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      a[i * 2 * n + j] = A[i * n + j];
      a[i * 2 * n + n + j] = (i == j) ? 1.0 : 0.0;
    }
  for (int c = 0; c < n; ++c) {
    int piv = c;
    for (int r = c + 1; r < n; ++r)
      if (fabs(a[r * 2 * n + c]) > fabs(a[piv * 2 * n + c])) piv = r;
    if (piv != c)
      for (int j = 0; j < 2 * n; ++j) {
        double t = a[c * 2 * n + j]; a[c * 2 * n + j] = a[piv * 2 * n + j]; a[piv * 2 * n + j] = t;
      }
    double d = 1.0 / a[c * 2 * n + c];
    for (int j = 0; j < 2 * n; ++j) a[c * 2 * n + j] *= d;
    for (int r = 0; r < n; ++r)
      if (r != c) {
        double f = a[r * 2 * n + c];
        for (int j = 0; j < 2 * n; ++j) a[r * 2 * n + j] -= f * a[c * 2 * n + j];
      }
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) Ai[i * n + j] = a[i * 2 * n + n + j];
}

static void mat_mul(int m, int k, int n, const double *A, const double *B, double *C)
{
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < n; ++j) {
      double s = 0;
      for (int l = 0; l < k; ++l) s += A[i * k + l] * B[l * n + j];
      C[i * n + j] = s;
    }
}

/* fe_time.h:157-305 */
int stfo_time_weights_wave(int type, int r, double tau, int nsteps, double *Alpha_lhs,
                           double *Beta_lhs, double *rhs_uK, double *rhs_uM, double *rhs_vM)
{
  double A[MAXN * MAXN], B[MAXN * MAXN], G[MAXN], Z[MAXN], t2[MAXN], t3[MAXN];
  const int nt = weights_1(type, r, tau, A, B, G, Z, t2, t3);
  if (nt < 0) return -1;
  const int nb = nt * nsteps;
  double Ai[MAXN * MAXN], T[MAXN * MAXN], BAB[MAXN * MAXN], BAG[MAXN], GAG[MAXN],
    GAB[MAXN * MAXN];
  mat_inv(nt, A, Ai);
  mat_mul(nt, nt, nt, B, Ai, T);    /* B A^-1 */
  mat_mul(nt, nt, nt, T, B, BAB);   /* B A^-1 B */
  mat_mul(nt, nt, 1, T, G, BAG);    /* B A^-1 Gamma */
  const double all = A[(nt - 1) * nt + nt - 1];
  const double gxai = G[nt - 1] / all;
  for (int i = 0; i < nt; ++i) GAG[i] = G[i] * gxai;
  for (int i = 0; i < nt; ++i)
    for (int j = 0; j < nt; ++j) GAB[i * nt + j] = G[i] * B[(nt - 1) * nt + j] / all;

  memset(Alpha_lhs, 0, sizeof(double) * nb * nb);
  memset(Beta_lhs, 0, sizeof(double) * nb * nb);
  memset(rhs_uK, 0, sizeof(double) * nb);
  memset(rhs_uM, 0, sizeof(double) * nb);
  memset(rhs_vM, 0, sizeof(double) * nb);
#define AL(i, j) Alpha_lhs[(i) * nb + (j)]
#define BL(i, j) Beta_lhs[(i) * nb + (j)]
  if (type == 0) {
    double BAZ[MAXN], ZmBAG[MAXN], ZmBAB[MAXN * MAXN];
    mat_mul(nt, nt, 1, T, Z, BAZ);
    for (int i = 0; i < nt; ++i) ZmBAG[i] = Z[i] - BAG[i];
    for (int i = 0; i < nt; ++i)
      for (int j = 0; j < nt; ++j) ZmBAB[i * nt + j] = ZmBAG[i] * B[(nt - 1) * nt + j] / all;
    const double zxai = Z[nt - 1] / all;
    for (int it = 0; it < nsteps; ++it)
      for (int jt = 0; jt <= it; ++jt)
        for (int i = 0; i < nt; ++i) {
          if (it == 0 && jt == 0) {
            rhs_uK[i] = G[i]; rhs_uM[i] = BAZ[i]; rhs_vM[i] = ZmBAG[i];
          } else if (jt == 0) {
            rhs_uM[i + it * nt] = -zxai * pow(gxai, it - 1) * ZmBAG[i];
            rhs_vM[i + it * nt] = pow(gxai, it) * ZmBAG[i];
          }
          if (it == jt + 1) {
            AL(i + it * nt, nt - 1 + jt * nt) = -G[i];
            BL(i + it * nt, nt - 1 + jt * nt) = -BAZ[i];
          }
          if (it == jt)
            for (int j = 0; j < nt; ++j) {
              AL(i + it * nt, j + it * nt) = A[i * nt + j];
              BL(i + it * nt, j + it * nt) = BAB[i * nt + j];
            }
          else
            for (int j = 0; j < nt; ++j)
              BL(i + it * nt, j + jt * nt) +=
                -pow(gxai, it - jt - 1) * ZmBAB[i * nt + j] +
                ((it > 1 && it - 1 > jt && j == nt - 1) ?
                   pow(gxai, it - jt - 2) * zxai * ZmBAG[i] : 0.0);
        }
  } else {
    for (int it = 0; it < nsteps; ++it)
      for (int i = 0; i < nt; ++i) {
        if (it == 0) { rhs_uM[i] = BAG[i]; rhs_vM[i] = G[i]; }
        if (it == 1) rhs_uM[nt + i] = -GAG[i];
        if (it < nsteps - 1)
          for (int j = 0; j < nt; ++j)
            BL(j + (it + 1) * nt, i + it * nt) =
              -GAB[j * nt + i] - (i == nt - 1 ? BAG[j] : 0.0);
        if (it < nsteps - 2 && i == nt - 1)
          for (int j = 0; j < nt; ++j) BL(j + (it + 2) * nt, i + it * nt) = GAG[j];
        for (int j = 0; j < nt; ++j) {
          AL(i + it * nt, j + it * nt) = A[i * nt + j];
          BL(i + it * nt, j + it * nt) = BAB[i * nt + j];
        }
      }
  }
#undef AL
#undef BL
  return nb;
}
