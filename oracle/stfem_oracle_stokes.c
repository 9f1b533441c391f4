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
 * Only the cell loop (LoopType::Cell, operators.h:1228-1229: no weak boundary ids, delta0 = 0) is
 * restated; the Nitsche / outflow / CIP face terms (operators.h:1577-1751) are not.
 * Velocity: FE_Q(pu)^3, pressure: FE_Q(pu-1), QGauss(pu+1), MappingQ1, homogeneous Dirichlet
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
  const long Nu = (long)ndu[0] * ndu[1] * ndu[2], Np = (long)ndp[0] * ndp[1] * ndp[2];
  double xq[MAXN], wq[MAXN], Su[MAXN * MAXN], Du[MAXN * MAXN], Sp[MAXN * MAXN], Dp[MAXN * MAXN];
  stfo_gauss(nq, xq, wq);
  stfo_shape_tables(pu, nq, Su, Du); /* S[q*(pu+1)+a] */
  stfo_shape_tables(pp, nq, Sp, Dp);
  if (!add) {
    memset(out_u, 0, sizeof(double) * 3 * Nu);
    memset(out_p, 0, sizeof(double) * Np);
  }
  const int nvx = nc[0] + 1, nvy = nc[1] + 1;
  const int nun = nu1 * nu1 * nu1, npn = np1 * np1 * np1, nqq = nq * nq * nq;
  double *ul = malloc(sizeof(double) * 3 * nun), *pl = malloc(sizeof(double) * npn);
  double *ru = malloc(sizeof(double) * 3 * nun), *rp = malloc(sizeof(double) * npn);
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
        for (int c = 0; c < np1; ++c)
          for (int b = 0; b < np1; ++b)
            for (int a = 0; a < np1; ++a)
              out_p[(pp * cx + a) + (long)ndp[0] * ((pp * cy + b) + (long)ndp[1] * (pp * cz + c))] += rp[a + np1 * (b + np1 * c)];
      }
  free(ul); free(pl); free(ru); free(rp);
  return 0;
}
