/*
 * stfem_oracle_stokes.c -- CPU ORACLE of the Stokes two-field cell operator (test
 * infrastructure, NOT product code; see stfem_oracle.h).
 *
 * Restates, in the reference's structure (gather -> evaluate -> quadrature loop -> integrate ->
 * scatter, one cell at a time):
 *   include/operators.h:1501-1523  StokesMatrixFreeOperator::do_cell_integral_range
 *   include/operators.h:1525-1575  do_cell_integral_local, OperatorMode::none:
 *        pressure.submit_value(div u);  velocity.submit_gradient(nu * grad u - p I)
 *   include/operators.h:1013-1018, 1135-1173  the vector mass operator (MatrixFreeOperator with
 *        n_components = dim, mass_scaling 1) used for the d/dt u term
 * plus (round 3) the boundary-face loop of LoopType::Full for the LINEAR operator:
 *   include/operators.h:1640-1660, 1662-1741  do_boundary_integral_range / do_boundary_face_integral_local:
 *        weak (Nitsche) faces: v <- -nu grad u n + p n + gamma1/h u + gamma2/h n (u.n),  dv/dn <- -nu u,  q <- -u.n
 *        with gamma1 = nu penalty1, gamma2 = penalty2 (1220-1221), h = (face area)^(1/(dim-1)) (184-209);
 *        outflow faces contribute nothing in the linear case (bfp carries a factor 0.0, dn = 0);
 *   include/operators.h:1898-1940  StokesNitscheMatrixFreeOperator::do_boundary_integral_range (the Dirichlet data g).
 * Not restated: the CIP interior-face term (delta0 != 0, 1603-1638; nonlinear in the velocity) and the convection modes.
 * Velocity: FE_Q(pu)^3, pressure: FE_Q(pu-1) or (pspace = 1) FE_DGP(pu-1) as tests/tp_03stokes.cc:83-86 selects it with
 * dGPressure (deal.II's basis: the Legendre polynomials on [0,1], orthonormal, complete degree pu-1, ordered x fastest:
 * for pu = 2: 1, sqrt 3 (2 xi - 1), sqrt 3 (2 eta - 1), sqrt 3 (2 zeta - 1); DoFs cell by cell), QGauss(pu+1), MappingQ1, homogeneous Dirichlet
 * constraints on the velocity only.  DoF layout: velocity = 3 component arrays of the scalar
 * FE_Q(pu) numbering (component-major), pressure = scalar FE_Q(pu-1) numbering, lexicographic.
 * Parity: pinned by an independent dense numpy assembly (tests/golden/make_golden.py); the
 * reference ships no golden vector of this operator ("parity unpinned" against its binary).
 */
#include "stfem_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAXN 5 /* nodes per direction (pu <= 4) */

static void trilinear_jac(const double v[8][3], double x, double y, double z, double J[3][3])
{
  const double fx[2] = {1 - x, x}, fy[2] = {1 - y, y}, fz[2] = {1 - z, z}, dd[2] = {-1, 1};
  memset(J, 0, 9 * sizeof(double));
  for (int k = 0; k < 2; ++k)
    for (int j = 0; j < 2; ++j)
      for (int i = 0; i < 2; ++i) {
        const double *X = v[i + 2 * j + 4 * k];
        for (int d = 0; d < 3; ++d) {
          J[d][0] += X[d] * dd[i] * fy[j] * fz[k];
          J[d][1] += X[d] * fx[i] * dd[j] * fz[k];
          J[d][2] += X[d] * fx[i] * fy[j] * dd[k];
        }
      }
}

static double inv3(const double J[3][3], double Ji[3][3])
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
  return det; /* Ji[e][d] = d xi_e / d x_d */
}

long stfo_stokes_n_velocity(const int nc[3], int pu)
{
  return (long)(pu * nc[0] + 1) * (pu * nc[1] + 1) * (pu * nc[2] + 1);
}
long stfo_stokes_n_pressure(const int nc[3], int pu)
{
  const int pp = pu - 1;
  return (long)(pp * nc[0] + 1) * (pp * nc[1] + 1) * (pp * nc[2] + 1);
}

/* ---- pressure space: 0 = FE_Q(pu - 1) (continuous, nodal), 1 = FE_DGP(pu - 1) (discontinuous, Legendre basis per cell) ---- */
static int g_pspace = 0;
void stfo_stokes_set_pressure_space(int pspace) { g_pspace = pspace; }
static int dgp_n(int pp) { return (pp + 1) * (pp + 2) * (pp + 3) / 6; }
long stfo_stokes_n_pressure_space(const int nc[3], int pu, int pspace)
{
  if (!pspace) return stfo_stokes_n_pressure(nc, pu);
  return (long)nc[0] * nc[1] * nc[2] * dgp_n(pu - 1);
}
/* orthonormal Legendre polynomial of degree n on [0, 1] */
static double legendre01(int n, double x)
{
  const double t = 2 * x - 1;
  double p0 = 1, p1 = t;
  if (n == 0) return 1.0;
  for (int k = 2; k <= n; ++k) {
    const double pk = ((2 * k - 1) * t * p1 - (k - 1) * p0) / k;
    p0 = p1; p1 = pk;
  }
  return sqrt(2.0 * n + 1.0) * p1;
}
/* values of the FE_DGP(pp) basis at a reference point: PolynomialSpace order (total degree <= pp, x index fastest) */
static int dgp_values(int pp, double x, double y, double z, double *out)
{
  int n = 0;
  for (int k = 0; k <= pp; ++k)
    for (int j = 0; j + k <= pp; ++j)
      for (int i = 0; i + j + k <= pp; ++i) out[n++] = legendre01(i, x) * legendre01(j, y) * legendre01(k, z);
  return n;
}

/* out_u (+)= wK * (nu K u - B^T p) + wM * M u ;  out_p (+)= wK * B u      (B u = (div u, q))
 * u, out_u: 3 * Nu doubles (component-major); p, out_p: Np doubles. */
int stfo_stokes_apply(const int nc[3], const double *vertices, int pu, int dirichlet_mask, double nu,
                      double wK, double wM, const double *u, const double *p, double *out_u,
                      double *out_p, int add)
{
  if (pu < 2 || pu > 4) return -1;
  const int pp = pu - 1, nu1 = pu + 1, np1 = pp + 1, nq = pu + 1;
  const int ndu[3] = {pu * nc[0] + 1, pu * nc[1] + 1, pu * nc[2] + 1};
  const int ndp[3] = {pp * nc[0] + 1, pp * nc[1] + 1, pp * nc[2] + 1};
  const long Nu = (long)ndu[0] * ndu[1] * ndu[2], Np = stfo_stokes_n_pressure_space(nc, pu, g_pspace);
  const int ndg = dgp_n(pp);
  double xq[MAXN], wq[MAXN], Su[MAXN * MAXN], Du[MAXN * MAXN], Sp[MAXN * MAXN], Dp[MAXN * MAXN];
  stfo_gauss(nq, xq, wq);
  stfo_shape_tables(pu, nq, Su, Du); /* S[q*(pu+1)+a] */
  stfo_shape_tables(pp, nq, Sp, Dp);
  if (!add) {
    memset(out_u, 0, sizeof(double) * 3 * Nu);
    memset(out_p, 0, sizeof(double) * Np);
  }
  const int nvx = nc[0] + 1, nvy = nc[1] + 1;
  const int nun = nu1 * nu1 * nu1, npn = g_pspace ? ndg : np1 * np1 * np1, nqq = nq * nq * nq;
  double *ul = malloc(sizeof(double) * 3 * nun), *pl = malloc(sizeof(double) * npn);
  double *ru = malloc(sizeof(double) * 3 * nun), *rp = malloc(sizeof(double) * npn);
  double pb[64];
  for (int cz = 0; cz < nc[2]; ++cz)
    for (int cy = 0; cy < nc[1]; ++cy)
      for (int cx = 0; cx < nc[0]; ++cx) {
        double v[8][3];
        for (int k = 0; k < 2; ++k)
          for (int j = 0; j < 2; ++j)
            for (int i = 0; i < 2; ++i)
              for (int d = 0; d < 3; ++d)
                v[i + 2 * j + 4 * k][d] = vertices[3 * ((cx + i) + (long)nvx * ((cy + j) + (long)nvy * (cz + k))) + d];
        /* gather (read_dof_values: constrained entries read as 0) */
        for (int c = 0; c < nu1; ++c)
          for (int b = 0; b < nu1; ++b)
            for (int a = 0; a < nu1; ++a) {
              const int ix = pu * cx + a, iy = pu * cy + b, iz = pu * cz + c;
              const int con = ((dirichlet_mask & 1) && ix == 0) || ((dirichlet_mask & 2) && ix == ndu[0] - 1) ||
                              ((dirichlet_mask & 4) && iy == 0) || ((dirichlet_mask & 8) && iy == ndu[1] - 1) ||
                              ((dirichlet_mask & 16) && iz == 0) || ((dirichlet_mask & 32) && iz == ndu[2] - 1);
              const long g = ix + (long)ndu[0] * (iy + (long)ndu[1] * iz);
              for (int comp = 0; comp < 3; ++comp) ul[comp * nun + a + nu1 * (b + nu1 * c)] = con ? 0.0 : u[comp * Nu + g];
            }
        if (g_pspace) {
          const long cell = cx + (long)nc[0] * (cy + (long)nc[1] * cz);
          for (int j = 0; j < ndg; ++j) pl[j] = p[cell * ndg + j];
        } else
        for (int c = 0; c < np1; ++c)
          for (int b = 0; b < np1; ++b)
            for (int a = 0; a < np1; ++a)
              pl[a + np1 * (b + np1 * c)] = p[(pp * cx + a) + (long)ndp[0] * ((pp * cy + b) + (long)ndp[1] * (pp * cz + c))];
        memset(ru, 0, sizeof(double) * 3 * nun);
        memset(rp, 0, sizeof(double) * npn);
        for (int q = 0; q < nqq; ++q) {
          const int qx = q % nq, qy = (q / nq) % nq, qz = q / (nq * nq);
          double J[3][3], Ji[3][3];
          trilinear_jac(v, xq[qx], xq[qy], xq[qz], J);
          const double JxW = inv3(J, Ji) * wq[qx] * wq[qy] * wq[qz];
          /* evaluate: reference gradient and value of the velocity, value of the pressure */
          double gref[3][3] = {{0}}, uval[3] = {0, 0, 0}, pval = 0;
          for (int c = 0; c < nu1; ++c)
            for (int b = 0; b < nu1; ++b)
              for (int a = 0; a < nu1; ++a) {
                const int n = a + nu1 * (b + nu1 * c);
                const double sx = Su[qx * nu1 + a], sy = Su[qy * nu1 + b], sz = Su[qz * nu1 + c];
                const double dx = Du[qx * nu1 + a] * sy * sz, dy = sx * Du[qy * nu1 + b] * sz,
                             dz = sx * sy * Du[qz * nu1 + c], val = sx * sy * sz;
                for (int comp = 0; comp < 3; ++comp) {
                  const double w = ul[comp * nun + n];
                  gref[comp][0] += w * dx; gref[comp][1] += w * dy; gref[comp][2] += w * dz;
                  uval[comp] += w * val;
                }
              }
          if (g_pspace) {
            dgp_values(pp, xq[qx], xq[qy], xq[qz], pb);
            for (int j = 0; j < ndg; ++j) pval += pl[j] * pb[j];
          } else
          for (int c = 0; c < np1; ++c)
            for (int b = 0; b < np1; ++b)
              for (int a = 0; a < np1; ++a)
                pval += pl[a + np1 * (b + np1 * c)] * Sp[qx * np1 + a] * Sp[qy * np1 + b] * Sp[qz * np1 + c];
          /* get_gradient: grad[comp][d] = sum_e gref[comp][e] * dxi_e/dx_d */
          double grad[3][3], divu = 0;
          for (int comp = 0; comp < 3; ++comp)
            for (int d = 0; d < 3; ++d)
              grad[comp][d] = gref[comp][0] * Ji[0][d] + gref[comp][1] * Ji[1][d] + gref[comp][2] * Ji[2][d];
          divu = grad[0][0] + grad[1][1] + grad[2][2];
          /* operators.h:1547-1553, 1570: submit */
          double F[3][3];
          for (int comp = 0; comp < 3; ++comp)
            for (int d = 0; d < 3; ++d) F[comp][d] = wK * (nu * grad[comp][d] - (comp == d ? pval : 0.0)) * JxW;
          const double dq = wK * divu * JxW;
          /* integrate: test gradients back to reference coordinates, test values */
          double Fref[3][3];
          for (int comp = 0; comp < 3; ++comp)
            for (int e = 0; e < 3; ++e)
              Fref[comp][e] = Ji[e][0] * F[comp][0] + Ji[e][1] * F[comp][1] + Ji[e][2] * F[comp][2];
          for (int c = 0; c < nu1; ++c)
            for (int b = 0; b < nu1; ++b)
              for (int a = 0; a < nu1; ++a) {
                const int n = a + nu1 * (b + nu1 * c);
                const double sx = Su[qx * nu1 + a], sy = Su[qy * nu1 + b], sz = Su[qz * nu1 + c];
                const double dx = Du[qx * nu1 + a] * sy * sz, dy = sx * Du[qy * nu1 + b] * sz,
                             dz = sx * sy * Du[qz * nu1 + c], val = sx * sy * sz;
                for (int comp = 0; comp < 3; ++comp)
                  ru[comp * nun + n] += dx * Fref[comp][0] + dy * Fref[comp][1] + dz * Fref[comp][2] +
                                        wM * val * uval[comp] * JxW;
              }
          if (g_pspace) {
            for (int j = 0; j < ndg; ++j) rp[j] += pb[j] * dq;
          } else
          for (int c = 0; c < np1; ++c)
            for (int b = 0; b < np1; ++b)
              for (int a = 0; a < np1; ++a)
                rp[a + np1 * (b + np1 * c)] += Sp[qx * np1 + a] * Sp[qy * np1 + b] * Sp[qz * np1 + c] * dq;
        }
        /* distribute_local_to_global: constrained rows are not written */
        for (int c = 0; c < nu1; ++c)
          for (int b = 0; b < nu1; ++b)
            for (int a = 0; a < nu1; ++a) {
              const int ix = pu * cx + a, iy = pu * cy + b, iz = pu * cz + c;
              const int con = ((dirichlet_mask & 1) && ix == 0) || ((dirichlet_mask & 2) && ix == ndu[0] - 1) ||
                              ((dirichlet_mask & 4) && iy == 0) || ((dirichlet_mask & 8) && iy == ndu[1] - 1) ||
                              ((dirichlet_mask & 16) && iz == 0) || ((dirichlet_mask & 32) && iz == ndu[2] - 1);
              if (con) continue;
              const long g = ix + (long)ndu[0] * (iy + (long)ndu[1] * iz);
              for (int comp = 0; comp < 3; ++comp) out_u[comp * Nu + g] += ru[comp * nun + a + nu1 * (b + nu1 * c)];
            }
        if (g_pspace) {
          const long cell = cx + (long)nc[0] * (cy + (long)nc[1] * cz);
          for (int j = 0; j < ndg; ++j) out_p[cell * ndg + j] += rp[j];
        } else
        for (int c = 0; c < np1; ++c)
          for (int b = 0; b < np1; ++b)
            for (int a = 0; a < np1; ++a)
              out_p[(pp * cx + a) + (long)ndp[0] * ((pp * cy + b) + (long)ndp[1] * (pp * cz + c))] += rp[a + np1 * (b + np1 * c)];
      }
  free(ul); free(pl); free(ru); free(rp);
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Boundary faces.  Face f = 2 d + s of the box: direction d, side s (0: lower, 1: upper) - deal.II's
 * boundary ids of a colorized hyper_rectangle.  Face quadrature: the cell rule's 1D Gauss points in the
 * two tangential directions t1 < t2, q = q1 + nq q2.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  double xyz[3], n[3], JxW;
  double Ji[3][3];
} face_point;

static void face_geometry(const double v[8][3], int d, int s, int nq, const double *xq, const double *wq,
                          face_point *fp, double *h)
{
  const int t1 = d == 0 ? 1 : 0, t2 = d == 2 ? 1 : 2;
  double area = 0;
  for (int q2 = 0; q2 < nq; ++q2)
    for (int q1 = 0; q1 < nq; ++q1) {
      double xi[3], J[3][3];
      xi[d] = s; xi[t1] = xq[q1]; xi[t2] = xq[q2];
      trilinear_jac(v, xi[0], xi[1], xi[2], J);
      face_point *P = fp + q1 + nq * q2;
      const double det = inv3(J, P->Ji);
      double m[3], len = 0;
      for (int k = 0; k < 3; ++k) { m[k] = (s ? 1.0 : -1.0) * P->Ji[d][k]; len += m[k] * m[k]; }
      len = sqrt(len);
      for (int k = 0; k < 3; ++k) P->n[k] = m[k] / len;
      P->JxW = fabs(det) * len * wq[q1] * wq[q2];
      area += P->JxW;
      const double fx[2] = {1 - xi[0], xi[0]}, fy[2] = {1 - xi[1], xi[1]}, fz[2] = {1 - xi[2], xi[2]};
      for (int c = 0; c < 3; ++c) P->xyz[c] = 0;
      for (int k = 0; k < 2; ++k)
        for (int j = 0; j < 2; ++j)
          for (int i = 0; i < 2; ++i)
            for (int c = 0; c < 3; ++c) P->xyz[c] += v[i + 2 * j + 4 * k][c] * fx[i] * fy[j] * fz[k];
    }
  *h = sqrt(area); /* get_h_face: area^(1/(dim-1)) */
}

static int is_con(int mask, const int ndu[3], int ix, int iy, int iz)
{
  return ((mask & 1) && ix == 0) || ((mask & 2) && ix == ndu[0] - 1) || ((mask & 4) && iy == 0) ||
         ((mask & 8) && iy == ndu[1] - 1) || ((mask & 16) && iz == 0) || ((mask & 32) && iz == ndu[2] - 1);
}

/* number of face quadrature points of the faces in weak_mask, in the order of stfo_stokes_face_points */
long stfo_stokes_n_face_points(const int nc[3], int pu, int weak_mask)
{
  const int nq = pu + 1;
  long n = 0;
  for (int f = 0; f < 6; ++f)
    if (weak_mask & (1 << f)) {
      const int d = f / 2, t1 = d == 0 ? 1 : 0, t2 = d == 2 ? 1 : 2;
      n += (long)nc[t1] * nc[t2] * nq * nq;
    }
  return n;
}

/* mode 0: out_xyz[npts][3] <- the points (faces ascending, cells of a face lexicographic with t1 fastest, q = q1 + nq q2);
 * mode 1: the weak-face terms of StokesMatrixFreeOperator::vmult, scaled by wK, ADDED to out_u / out_p;
 * mode 2: StokesNitscheMatrixFreeOperator::vmult: the terms of the Dirichlet data g[npts][3] ADDED to out_u / out_p. */
static int boundary_loop(int mode, const int nc[3], const double *vertices, int pu, int dirichlet_mask, int weak_mask,
                         double nu, double penalty1, double penalty2, double wK, const double *u, const double *p,
                         const double *g, double *out_u, double *out_p, double *out_xyz)
{
  if (pu < 2 || pu > 4) return -1;
  const int pp = pu - 1, nu1 = pu + 1, np1 = pp + 1, nq = pu + 1;
  const int ndu[3] = {pu * nc[0] + 1, pu * nc[1] + 1, pu * nc[2] + 1};
  const int ndp[3] = {pp * nc[0] + 1, pp * nc[1] + 1, pp * nc[2] + 1};
  const long Nu = (long)ndu[0] * ndu[1] * ndu[2];
  const double gamma1 = nu * penalty1, gamma2 = penalty2;
  double xq[MAXN], wq[MAXN], Su[MAXN * MAXN], Du[MAXN * MAXN], Sp[MAXN * MAXN], Dp[MAXN * MAXN];
  double Eu[2 * MAXN], EDu[2 * MAXN], Ep[2 * MAXN], EDp[2 * MAXN];
  const double ends[2] = {0.0, 1.0};
  stfo_gauss(nq, xq, wq);
  stfo_shape_tables(pu, nq, Su, Du);
  stfo_shape_tables(pp, nq, Sp, Dp);
  stfo_shape_tables_at(pu, 2, ends, Eu, EDu);
  stfo_shape_tables_at(pp, 2, ends, Ep, EDp);
  const int nvx = nc[0] + 1, nvy = nc[1] + 1;
  const int ndg = dgp_n(pp);
  const int nun = nu1 * nu1 * nu1, npn = g_pspace ? ndg : np1 * np1 * np1;
  double *ul = malloc(sizeof(double) * 3 * nun), *pl = malloc(sizeof(double) * npn);
  double *ru = malloc(sizeof(double) * 3 * nun), *rp = malloc(sizeof(double) * npn);
  double pb[64];
  face_point fp[MAXN * MAXN];
  long pt = 0;
  for (int f = 0; f < 6; ++f) {
    if (!(weak_mask & (1 << f))) continue;
    const int d = f / 2, s = f % 2, t1 = d == 0 ? 1 : 0, t2 = d == 2 ? 1 : 2;
    for (int c2 = 0; c2 < nc[t2]; ++c2)
      for (int c1 = 0; c1 < nc[t1]; ++c1) {
        int cc[3];
        cc[d] = s ? nc[d] - 1 : 0; cc[t1] = c1; cc[t2] = c2;
        const int cx = cc[0], cy = cc[1], cz = cc[2];
        double v[8][3], h;
        for (int k = 0; k < 2; ++k)
          for (int j = 0; j < 2; ++j)
            for (int i = 0; i < 2; ++i)
              for (int e = 0; e < 3; ++e)
                v[i + 2 * j + 4 * k][e] = vertices[3 * ((cx + i) + (long)nvx * ((cy + j) + (long)nvy * (cz + k))) + e];
        face_geometry(v, d, s, nq, xq, wq, fp, &h);
        if (mode == 0) {
          for (int q = 0; q < nq * nq; ++q, ++pt)
            for (int e = 0; e < 3; ++e) out_xyz[3 * pt + e] = fp[q].xyz[e];
          continue;
        }
        if (mode == 1) {
          for (int c = 0; c < nu1; ++c)
            for (int b = 0; b < nu1; ++b)
              for (int a = 0; a < nu1; ++a) {
                const int ix = pu * cx + a, iy = pu * cy + b, iz = pu * cz + c;
                const long gi = ix + (long)ndu[0] * (iy + (long)ndu[1] * iz);
                for (int comp = 0; comp < 3; ++comp)
                  ul[comp * nun + a + nu1 * (b + nu1 * c)] = is_con(dirichlet_mask, ndu, ix, iy, iz) ? 0.0 : u[comp * Nu + gi];
              }
          if (g_pspace) {
            const long cell = cx + (long)nc[0] * (cy + (long)nc[1] * cz);
            for (int j = 0; j < ndg; ++j) pl[j] = p[cell * ndg + j];
          } else
          for (int c = 0; c < np1; ++c)
            for (int b = 0; b < np1; ++b)
              for (int a = 0; a < np1; ++a)
                pl[a + np1 * (b + np1 * c)] = p[(pp * cx + a) + (long)ndp[0] * ((pp * cy + b) + (long)ndp[1] * (pp * cz + c))];
        }
        memset(ru, 0, sizeof(double) * 3 * nun);
        memset(rp, 0, sizeof(double) * npn);
        for (int q2 = 0; q2 < nq; ++q2)
          for (int q1 = 0; q1 < nq; ++q1, ++pt) {
            const face_point *P = fp + q1 + nq * q2;
            /* 1D tables of this point: direction d at the end point s, t1 / t2 at the Gauss points */
            const double *SU[3], *DU[3], *SP[3];
            SU[d] = Eu + s * nu1; DU[d] = EDu + s * nu1; SP[d] = Ep + s * np1;
            SU[t1] = Su + q1 * nu1; DU[t1] = Du + q1 * nu1; SP[t1] = Sp + q1 * np1;
            SU[t2] = Su + q2 * nu1; DU[t2] = Du + q2 * nu1; SP[t2] = Sp + q2 * np1;
            double val[3] = {0, 0, 0}, nd[3] = {0, 0, 0}, pq = 0;
            if (g_pspace) { /* the DGP basis at this face point (reference coordinates) */
              double xi[3];
              xi[d] = s; xi[t1] = xq[q1]; xi[t2] = xq[q2];
              dgp_values(pp, xi[0], xi[1], xi[2], pb);
            }
            if (mode == 1) {
              double gref[3][3] = {{0}}, uval[3] = {0, 0, 0}, pval = 0;
              for (int c = 0; c < nu1; ++c)
                for (int b = 0; b < nu1; ++b)
                  for (int a = 0; a < nu1; ++a) {
                    const int n = a + nu1 * (b + nu1 * c);
                    const double sx = SU[0][a], sy = SU[1][b], sz = SU[2][c];
                    const double dx = DU[0][a] * sy * sz, dy = sx * DU[1][b] * sz, dz = sx * sy * DU[2][c];
                    for (int comp = 0; comp < 3; ++comp) {
                      const double w = ul[comp * nun + n];
                      gref[comp][0] += w * dx; gref[comp][1] += w * dy; gref[comp][2] += w * dz;
                      uval[comp] += w * sx * sy * sz;
                    }
                  }
              if (g_pspace) {
                for (int j = 0; j < ndg; ++j) pval += pl[j] * pb[j];
              } else
              for (int c = 0; c < np1; ++c)
                for (int b = 0; b < np1; ++b)
                  for (int a = 0; a < np1; ++a) pval += pl[a + np1 * (b + np1 * c)] * SP[0][a] * SP[1][b] * SP[2][c];
              double un = 0, gn[3];
              for (int comp = 0; comp < 3; ++comp) {
                gn[comp] = 0;
                for (int k = 0; k < 3; ++k) {
                  const double gk = gref[comp][0] * P->Ji[0][k] + gref[comp][1] * P->Ji[1][k] + gref[comp][2] * P->Ji[2][k];
                  gn[comp] += gk * P->n[k];
                }
                un += uval[comp] * P->n[comp];
              }
              /* operators.h:1727-1739 */
              for (int comp = 0; comp < 3; ++comp) {
                val[comp] = wK * (-nu * gn[comp] + pval * P->n[comp] + (gamma1 / h) * uval[comp] + (gamma2 / h) * P->n[comp] * un) * P->JxW;
                nd[comp] = wK * (-nu * uval[comp]) * P->JxW;
              }
              pq = wK * (-un) * P->JxW;
            } else {
              /* operators.h:1921-1932 (linear) */
              const double *gq = g + 3 * pt;
              const double gnn = gq[0] * P->n[0] + gq[1] * P->n[1] + gq[2] * P->n[2];
              for (int comp = 0; comp < 3; ++comp) {
                val[comp] = ((gamma1 / h) * gq[comp] + (gamma2 / h) * P->n[comp] * gnn) * P->JxW;
                nd[comp] = (-nu * gq[comp]) * P->JxW;
              }
              pq = -gnn * P->JxW;
            }
            /* integrate: test values and test normal derivatives */
            for (int c = 0; c < nu1; ++c)
              for (int b = 0; b < nu1; ++b)
                for (int a = 0; a < nu1; ++a) {
                  const int n = a + nu1 * (b + nu1 * c);
                  const double sx = SU[0][a], sy = SU[1][b], sz = SU[2][c];
                  const double gr[3] = {DU[0][a] * sy * sz, sx * DU[1][b] * sz, sx * sy * DU[2][c]};
                  double dn = 0;
                  for (int k = 0; k < 3; ++k) dn += (gr[0] * P->Ji[0][k] + gr[1] * P->Ji[1][k] + gr[2] * P->Ji[2][k]) * P->n[k];
                  for (int comp = 0; comp < 3; ++comp) ru[comp * nun + n] += sx * sy * sz * val[comp] + dn * nd[comp];
                }
            if (g_pspace) {
              for (int j = 0; j < ndg; ++j) rp[j] += pb[j] * pq;
            } else
            for (int c = 0; c < np1; ++c)
              for (int b = 0; b < np1; ++b)
                for (int a = 0; a < np1; ++a) rp[a + np1 * (b + np1 * c)] += SP[0][a] * SP[1][b] * SP[2][c] * pq;
          }
        for (int c = 0; c < nu1; ++c)
          for (int b = 0; b < nu1; ++b)
            for (int a = 0; a < nu1; ++a) {
              const int ix = pu * cx + a, iy = pu * cy + b, iz = pu * cz + c;
              if (is_con(dirichlet_mask, ndu, ix, iy, iz)) continue;
              const long gi = ix + (long)ndu[0] * (iy + (long)ndu[1] * iz);
              for (int comp = 0; comp < 3; ++comp) out_u[comp * Nu + gi] += ru[comp * nun + a + nu1 * (b + nu1 * c)];
            }
        if (g_pspace) {
          const long cell = cx + (long)nc[0] * (cy + (long)nc[1] * cz);
          for (int j = 0; j < ndg; ++j) out_p[cell * ndg + j] += rp[j];
        } else
        for (int c = 0; c < np1; ++c)
          for (int b = 0; b < np1; ++b)
            for (int a = 0; a < np1; ++a)
              out_p[(pp * cx + a) + (long)ndp[0] * ((pp * cy + b) + (long)ndp[1] * (pp * cz + c))] += rp[a + np1 * (b + np1 * c)];
      }
  }
  free(ul); free(pl); free(ru); free(rp);
  return 0;
}

int stfo_stokes_face_points(const int nc[3], const double *vertices, int pu, int weak_mask, double *out_xyz)
{
  return boundary_loop(0, nc, vertices, pu, 0, weak_mask, 1.0, 0, 0, 0, NULL, NULL, NULL, NULL, NULL, out_xyz);
}
int stfo_stokes_boundary_apply(const int nc[3], const double *vertices, int pu, int dirichlet_mask, int weak_mask, double nu,
                               double penalty1, double penalty2, double wK, const double *u, const double *p, double *out_u,
                               double *out_p)
{
  return boundary_loop(1, nc, vertices, pu, dirichlet_mask, weak_mask, nu, penalty1, penalty2, wK, u, p, NULL, out_u, out_p, NULL);
}
int stfo_stokes_nitsche_rhs(const int nc[3], const double *vertices, int pu, int dirichlet_mask, int weak_mask, double nu,
                            double penalty1, double penalty2, const double *g_at_face_points, double *out_u, double *out_p)
{
  return boundary_loop(2, nc, vertices, pu, dirichlet_mask, weak_mask, nu, penalty1, penalty2, 1.0, NULL, NULL, g_at_face_points,
                       out_u, out_p, NULL);
}
