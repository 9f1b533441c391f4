// Stokes two-field cell operator on MI355X (BASELINE configs[4]).
//
// Replaces, for the cell loop (LoopType::Cell: no weak boundary ids, delta0 = 0):
//   StokesMatrixFreeOperator::do_cell_integral_range / do_cell_integral_local
//       (reference include/operators.h:1501-1575, OperatorMode::none):
//       pressure.submit_value(div u); velocity.submit_gradient(nu grad u - p I)
//   the vector mass operator behind d/dt u (MatrixFreeOperator<dim, dim, Number>, operators.h:1135-1173)
//   SystemMatrixStokes::vmult -> tensorproduct_eval (operators.h:696-700, 825-867): per source time
//       dof one K.vmult + scatter with Alpha and one M.vmult + scatter with Beta.
// Here ONE launch per source time dof evaluates the cell once and scatters
//       wKu_j * (nu K u - B^T p) + wM_j * M u   into every velocity destination block j,
//       wKp_j * (div u, q)                       into every pressure destination block j
// in eight launches, one per cell colour (cells of one colour share no DoF), with plain loads and
// stores: no atomics, no zeroing, bitwise reproducible.  No CPU fallback.
//
// Thread layout: one wave owns two cells (32 lanes each, 27 = 3^3 active).  Evaluation and
// integration are sum-factorised (three 1D stages each, see stokes_cell_kernel); the MappingQ1
// Jacobian is evaluated on the fly from the eight cell vertices (24 doubles per cell instead of a
// stored metric), or is a constant diagonal on axis-aligned boxes.
#include "../../include/stfem.h"
#include "host_tables.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <vector>

namespace {

constexpr int MAXOUT = 8;
constexpr int MAXSRC = 4; // sources / destination pairs of the fused form

struct StokesParams {
  const double *vertices; // device, (nc+1)^3 * 3
  int ncx, ncy, ncz;
  int ndu[3], ndp[3];
  long long Nu, Np;
  int dmask;
  double nu;
  const double *u, *p; // source (u may be read with p == nullptr: mass only)
  int nout;
  double *out_u[MAXOUT], *out_p[MAXOUT];
  double wKu[MAXOUT], wKp[MAXOUT], wM[MAXOUT];
  double Su[9], Du[9], Sp[6]; // [q*3+a], [q*3+a], [q*2+a]
  double xq[3], wq[3];
  // fused form (SystemMatrixStokes::vmult with up to MAXSRC source time dofs and up to MAXSRC destination pairs in ONE set of colour
  // launches): the cell is evaluated for every source in turn, the weighted results are summed in registers and scattered once
  int interleave;             // cell -> half-wave assignment (see the kernel)
  int nsrc;                   // 0 / 1: the single source u, p with the weights above
  const double *us[4], *ps[4];
  double fKu[4][4], fKp[4][4], fM[4][4]; // [destination pair][source]
  int colour;                 // this launch handles the cells with (cx & 1) + 2 (cy & 1) + 4 (cz & 1) == colour
  int store_u[MAXOUT], store_p[MAXOUT]; // 1: the first cell to touch a DoF (lowest colour) stores, the others add; 0: all add
  int cart;                   // axis-aligned uniform cells: constant diagonal Jacobian
  double hinv[3], detJ;       // 1 / h_d, hx hy hz
  // pressure space: 0 = FE_Q(1) on the vertex lattice, 1 = FE_DGP(1), the reference's dGPressure (tests/tp_03stokes.cc:83-86):
  // four DoFs per cell, deal.II's basis 1, l(xi), l(eta), l(zeta) with l(x) = sqrt 3 (2 x - 1), p[cell * 4 + j]
  int pdg;
  double l1q[3];              // l at the three Gauss points
};

__device__ __forceinline__ bool constrained_u(const StokesParams &prm, int ix, int iy, int iz)
{
  return ((prm.dmask & 1) && ix == 0) || ((prm.dmask & 2) && ix == prm.ndu[0] - 1) ||
         ((prm.dmask & 4) && iy == 0) || ((prm.dmask & 8) && iy == prm.ndu[1] - 1) ||
         ((prm.dmask & 16) && iz == 0) || ((prm.dmask & 32) && iz == prm.ndu[2] - 1);
}

// Orders the LDS traffic of one wave (a cell lives in one half of a wave: no workgroup barrier needed)
__device__ __forceinline__ void wave_fence()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 256 threads = 4 waves = 8 cells at a time; the workgroups walk over the cells.
// Sum-factorised: evaluation and integration are three 1D stages each (x, y, z), handed from lane to
// lane through two wave-private LDS regions per cell that alternate as source and destination.
//   evaluate : lane (a, b, c) = (q_x, n_y, n_z) -> (q_x, q_y, n_z) -> quadrature point (q_x, q_y, q_z)
//   integrate: lane (q_x, q_y, n_z) -> (q_x, n_y, n_z) -> velocity node (n_x, n_y, n_z); the eight
//              lanes with a, b, c < 2 also carry the pressure node (a, b, c)
// The lane's rows / columns of the 1D tables stay in registers for the whole kernel.
// CART: axis-aligned uniform cells (the context was created without vertices): constant diagonal Jacobian.
// FUSED: several sources per cell, weighted sums in registers, one scatter (prm.nsrc > 1; see StokesParams)
template <bool CART, bool FUSED>
__global__ __launch_bounds__(256) void stokes_cell_kernel(const StokesParams prm)
{
  constexpr int RX = 351, RY = 351; // doubles per cell of the two regions (largest stage: 13 x 27)
  __shared__ double smem[8 * (RX + RY)];
  __shared__ double tS[9], tD[9], tP[6]; // 1D tables [q*3+a], [q*3+a], [q*2+a]
  if (threadIdx.x < 9) { tS[threadIdx.x] = prm.Su[threadIdx.x]; tD[threadIdx.x] = prm.Du[threadIdx.x]; }
  if (threadIdx.x < 6) tP[threadIdx.x] = prm.Sp[threadIdx.x];
  __syncthreads();
  const int slot = threadIdx.x >> 5, t32 = threadIdx.x & 31;
  const bool lane27 = t32 < 27;
  const int t = lane27 ? t32 : 0;
  const int a = t % 3, b = (t / 3) % 3, c = t / 9;
  const int a1 = a < 2 ? a : 1, b1 = b < 2 ? b : 1, c1 = c < 2 ? c : 1; // (pressure stages: clamped, unused where >= 2)
  double *X = smem + slot * (RX + RY), *Y = X + RX;
  // evaluation: row of this lane's quadrature index; integration: column of this lane's node index
  double Sa[3], Da[3], Sb[3], Db[3], Sc[3], Dc[3], SaT[3], DaT[3], SbT[3], DbT[3], ScT[3], DcT[3];
  double Pa[2], Pb[2], Pc[2], PaT[3], PbT[3], PcT[3];
#pragma unroll
  for (int n = 0; n < 3; ++n) {
    Sa[n] = tS[a * 3 + n]; Da[n] = tD[a * 3 + n]; Sb[n] = tS[b * 3 + n]; Db[n] = tD[b * 3 + n];
    Sc[n] = tS[c * 3 + n]; Dc[n] = tD[c * 3 + n];
    SaT[n] = tS[n * 3 + a]; DaT[n] = tD[n * 3 + a]; SbT[n] = tS[n * 3 + b]; DbT[n] = tD[n * 3 + b];
    ScT[n] = tS[n * 3 + c]; DcT[n] = tD[n * 3 + c];
    PaT[n] = tP[n * 2 + a1]; PbT[n] = tP[n * 2 + b1]; PcT[n] = tP[n * 2 + c1];
  }
#pragma unroll
  for (int n = 0; n < 2; ++n) { Pa[n] = tP[a * 2 + n]; Pb[n] = tP[b * 2 + n]; Pc[n] = tP[c * 2 + n]; }
  const double wabc = prm.wq[a] * prm.wq[b] * prm.wq[c];
  // this lane also carries a pressure DoF of the cell: FE_Q(1) node (a, b, c), or FE_DGP(1) function t
  const bool pnode = prm.pdg ? t32 < 4 : (lane27 && a < 2 && b < 2 && c < 2);
  const int pslot = prm.pdg ? t32 : a + 2 * b + 4 * c; // its slot in the cell's pressure values X[81 ..]
  const double la = prm.l1q[a], lb = prm.l1q[b], lc = prm.l1q[c]; // DGP: the linear functions at this lane's quadrature point
  // the cells of one colour share no DoF: the eight colours run as eight launches, lowest first, and
  // scatter with plain loads and stores (no atomics, no zeroing of the destinations, deterministic)
  const int px = prm.colour & 1, py = (prm.colour >> 1) & 1, pz = prm.colour >> 2;
  const int ncxc = (prm.ncx - px + 1) / 2, ncyc = (prm.ncy - py + 1) / 2, nczc = (prm.ncz - pz + 1) / 2;
  const long long ncells = (long long)ncxc * ncyc * nczc;

  // every half-wave walks through its own contiguous run of cells: cells sharing nodes are handled one
  // after the other by the same lanes instead of at the same time by neighbouring ones (their atomics
  // on the shared nodes would serialise in L2)
  const long long nhalf = (long long)gridDim.x * 8, run = (ncells + nhalf - 1) / nhalf;
  // STRIDE = 1: every half-wave walks its own contiguous run of cells; STRIDE = 8: the eight half-waves of the workgroup take eight
  // consecutive cells of the workgroup's run at a time (their rows are 32 bytes apart: denser sectors per gather / scatter instruction)
  const long long wg_first = (long long)blockIdx.x * 8 * run, wg_end = wg_first + 8 * run;
  const int STRIDE = prm.interleave ? 8 : 1;
  const long long first = prm.interleave ? wg_first + slot : ((long long)blockIdx.x * 8 + slot) * run;
  // the DoFs of a cell are fetched while the previous cell is being computed
  struct CellIds {
    int cx, cy, cz;
    bool ok, con;
    long long gu, gp;
  };
  auto ids = [&](long long cell) {
    CellIds q;
    q.ok = cell < ncells && (prm.interleave ? cell < wg_end : cell < first + run);
    const long long cc = q.ok ? cell : 0;
    q.cx = 2 * int(cc % ncxc) + px; q.cy = 2 * int((cc / ncxc) % ncyc) + py; q.cz = 2 * int(cc / ((long long)ncxc * ncyc)) + pz;
    const int ix = 2 * q.cx + a, iy = 2 * q.cy + b, iz = 2 * q.cz + c;
    q.con = constrained_u(prm, ix, iy, iz);
    q.gu = ix + (long long)prm.ndu[0] * (iy + (long long)prm.ndu[1] * iz);
    q.gp = prm.pdg ? (q.cx + (long long)prm.ncx * (q.cy + (long long)prm.ncy * q.cz)) * 4 + (t32 & 3)
                   : (q.cx + a1) + (long long)prm.ndp[0] * ((q.cy + b1) + (long long)prm.ndp[1] * (q.cz + c1));
    return q;
  };
  double un[3] = {0, 0, 0}, pn = 0.0;
  const int nsrc = FUSED ? prm.nsrc : 1;
  auto fetch = [&](const CellIds &q, int s) { // read_dof_values: constrained velocity entries read as 0
    const double *us = FUSED ? prm.us[s] : prm.u, *ps = FUSED ? prm.ps[s] : prm.p;
    if (q.ok && lane27 && !q.con) {
#pragma unroll
      for (int comp = 0; comp < 3; ++comp) un[comp] = us[comp * prm.Nu + q.gu];
    } else {
      un[0] = un[1] = un[2] = 0.0;
    }
    pn = (q.ok && pnode && ps) ? ps[q.gp] : 0.0;
  };
  CellIds nxt = ids(first);
  fetch(nxt, 0);
  double accU[FUSED ? MAXSRC : 1][3], accP[FUSED ? MAXSRC : 1]; // fused form: sums over the sources, per destination pair
#pragma unroll
  for (int o = 0; o < (FUSED ? MAXSRC : 1); ++o) accU[o][0] = accU[o][1] = accU[o][2] = accP[o] = 0.0;
  for (long long it2 = 0; it2 < run * nsrc; ++it2) {
    const long long it = it2 / nsrc;
    const int src = int(it2 - it * nsrc);
    const CellIds cur = nxt;
    const bool cell_ok = cur.ok;
    const int cx = cur.cx, cy = cur.cy, cz = cur.cz;
    const bool con = cur.con;
    const long long gu = cur.gu, gp = cur.gp;
    const bool active = cell_ok && lane27;
    (void)cx; (void)cy; (void)cz;

    // ---- gather: X = u[3][27], p[8]
    if (lane27) {
#pragma unroll
      for (int comp = 0; comp < 3; ++comp) X[comp * 27 + t] = un[comp];
    }
    if (pnode) X[81 + pslot] = pn;
    nxt = ids(first + STRIDE * ((it2 + 1) / nsrc));
    fetch(nxt, int((it2 + 1) % nsrc));
    wave_fence();
    double pdgv[4] = {0, 0, 0, 0};
    if (prm.pdg) {
#pragma unroll
      for (int j = 0; j < 4; ++j) pdgv[j] = X[81 + j];
    }

    // ---- evaluate, x: (n_x, n_y, n_z) -> (q_x, n_y, n_z): values and x derivatives -> Y
#pragma unroll
    for (int comp = 0; comp < 3; ++comp) {
      const double *u = X + comp * 27 + 3 * b + 9 * c;
      const double u0 = u[0], u1 = u[1], u2 = u[2];
      Y[(comp * 2) * 27 + t] = fma(Sa[2], u2, fma(Sa[1], u1, Sa[0] * u0));
      Y[(comp * 2 + 1) * 27 + t] = fma(Da[2], u2, fma(Da[1], u1, Da[0] * u0));
    }
    Y[162 + t] = fma(Pa[1], X[81 + 1 + 2 * b1 + 4 * c1], Pa[0] * X[81 + 2 * b1 + 4 * c1]); // pressure (n_y, n_z < 2)
    wave_fence();
    // ---- y: -> (q_x, q_y, n_z): value, d/dx, d/dy -> X
#pragma unroll
    for (int comp = 0; comp < 3; ++comp) {
      const double *v = Y + (comp * 2) * 27 + a + 9 * c, *d = v + 27;
      const double v0 = v[0], v1 = v[3], v2 = v[6], d0 = d[0], d1 = d[3], d2 = d[6];
      X[(comp * 3) * 27 + t] = fma(Sb[2], v2, fma(Sb[1], v1, Sb[0] * v0));
      X[(comp * 3 + 1) * 27 + t] = fma(Sb[2], d2, fma(Sb[1], d1, Sb[0] * d0));
      X[(comp * 3 + 2) * 27 + t] = fma(Db[2], v2, fma(Db[1], v1, Db[0] * v0));
    }
    X[243 + t] = fma(Pb[1], Y[162 + a + 3 + 9 * c1], Pb[0] * Y[162 + a + 9 * c1]);
    wave_fence();
    // ---- z: -> quadrature point (a, b, c): value and reference gradient in registers
    double uval[3], gref[3][3];
#pragma unroll
    for (int comp = 0; comp < 3; ++comp) {
      const double *v = X + (comp * 3) * 27 + a + 3 * b, *dx = v + 27, *dy = v + 54;
      const double v0 = v[0], v1 = v[9], v2 = v[18];
      uval[comp] = fma(Sc[2], v2, fma(Sc[1], v1, Sc[0] * v0));
      gref[comp][0] = fma(Sc[2], dx[18], fma(Sc[1], dx[9], Sc[0] * dx[0]));
      gref[comp][1] = fma(Sc[2], dy[18], fma(Sc[1], dy[9], Sc[0] * dy[0]));
      gref[comp][2] = fma(Dc[2], v2, fma(Dc[1], v1, Dc[0] * v0));
    }
    double pval = fma(Pc[1], X[243 + a + 3 * b + 9], Pc[0] * X[243 + a + 3 * b]);
    if (prm.pdg) pval = pdgv[0] + la * pdgv[1] + lb * pdgv[2] + lc * pdgv[3];

    // ---- quadrature-point operation (operators.h:1547-1553, 1570; weights applied at scatter time) -> Y
    if (CART) {
      const double JxW = prm.detJ * wabc;
#pragma unroll
      for (int comp = 0; comp < 3; ++comp) {
#pragma unroll
        for (int e = 0; e < 3; ++e) {
          const double g = gref[comp][e] * prm.hinv[e];
          Y[(comp * 3 + e) * 27 + t] = (prm.nu * g - (comp == e ? pval : 0.0)) * JxW * prm.hinv[e];
        }
        Y[(10 + comp) * 27 + t] = uval[comp] * JxW;
      }
      Y[9 * 27 + t] = (gref[0][0] * prm.hinv[0] + gref[1][1] * prm.hinv[1] + gref[2][2] * prm.hinv[2]) * JxW;
    } else {
      const double x = prm.xq[a], y = prm.xq[b], z = prm.xq[c];
      const double fx[2] = {1 - x, x}, fy[2] = {1 - y, y}, fz[2] = {1 - z, z}, dd[2] = {-1.0, 1.0};
      double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
      const long long nvx = prm.ncx + 1, nvy = prm.ncy + 1;
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const double *V = prm.vertices + 3 * ((cx + i) + nvx * ((cy + j) + nvy * (long long)(cz + k)));
#pragma unroll
            for (int d = 0; d < 3; ++d) {
              const double Vd = V[d];
              J[d][0] += Vd * dd[i] * fy[j] * fz[k];
              J[d][1] += Vd * fx[i] * dd[j] * fz[k];
              J[d][2] += Vd * fx[i] * fy[j] * dd[k];
            }
          }
      const double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
                         J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
      const double id = 1.0 / det;
      double Ji[3][3];
      Ji[0][0] = (J[1][1] * J[2][2] - J[1][2] * J[2][1]) * id;
      Ji[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id;
      Ji[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
      Ji[1][0] = (J[1][2] * J[2][0] - J[1][0] * J[2][2]) * id;
      Ji[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id;
      Ji[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
      Ji[2][0] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) * id;
      Ji[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id;
      Ji[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
      const double JxW = det * wabc;
      double divu = 0.0;
#pragma unroll
      for (int comp = 0; comp < 3; ++comp) {
        double F[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const double g = gref[comp][0] * Ji[0][d] + gref[comp][1] * Ji[1][d] + gref[comp][2] * Ji[2][d];
          if (comp == d) divu += g;
          F[d] = (prm.nu * g - (comp == d ? pval : 0.0)) * JxW;
        }
#pragma unroll
        for (int e = 0; e < 3; ++e) Y[(comp * 3 + e) * 27 + t] = Ji[e][0] * F[0] + Ji[e][1] * F[1] + Ji[e][2] * F[2];
        Y[(10 + comp) * 27 + t] = uval[comp] * JxW;
      }
      Y[9 * 27 + t] = divu * JxW;
    }
    wave_fence();
    double rPdg = 0.0;
    if (prm.pdg && t32 < 4) { // (q_j, div u): the cell's own four test functions, summed over the 27 quadrature points
      const double *fd = Y + 9 * 27;
      for (int q = 0; q < 27; ++q) {
        const int qa = q % 3, qb = (q / 3) % 3, qc = q / 9;
        const double l = t32 == 0 ? 1.0 : prm.l1q[t32 == 1 ? qa : (t32 == 2 ? qb : qc)];
        rPdg = fma(l, fd[q], rPdg);
      }
    }

    // ---- integrate, z: quadrature point -> (q_x, q_y, n_z) -> X
#pragma unroll
    for (int comp = 0; comp < 3; ++comp) {
      const double *f0 = Y + (comp * 3) * 27 + a + 3 * b, *f1 = f0 + 27, *f2 = f0 + 54, *fm = Y + (10 + comp) * 27 + a + 3 * b;
      X[(comp * 4) * 27 + t] = fma(ScT[2], f0[18], fma(ScT[1], f0[9], ScT[0] * f0[0]));
      X[(comp * 4 + 1) * 27 + t] = fma(ScT[2], f1[18], fma(ScT[1], f1[9], ScT[0] * f1[0]));
      X[(comp * 4 + 2) * 27 + t] = fma(DcT[2], f2[18], fma(DcT[1], f2[9], DcT[0] * f2[0]));
      X[(comp * 4 + 3) * 27 + t] = fma(ScT[2], fm[18], fma(ScT[1], fm[9], ScT[0] * fm[0]));
    }
    {
      const double *fd = Y + 9 * 27 + a + 3 * b;
      X[324 + t] = fma(PcT[2], fd[18], fma(PcT[1], fd[9], PcT[0] * fd[0]));
    }
    wave_fence();
    // ---- y: -> (q_x, n_y, n_z) -> Y
#pragma unroll
    for (int comp = 0; comp < 3; ++comp) {
      const double *g0 = X + (comp * 4) * 27 + a + 9 * c, *g1 = g0 + 27, *g2 = g0 + 54, *gm = g0 + 81;
      Y[(comp * 3) * 27 + t] = fma(SbT[2], g0[6], fma(SbT[1], g0[3], SbT[0] * g0[0]));
      Y[(comp * 3 + 1) * 27 + t] = fma(DbT[2], g1[6], fma(DbT[1], g1[3], DbT[0] * g1[0])) +
                                   fma(SbT[2], g2[6], fma(SbT[1], g2[3], SbT[0] * g2[0]));
      Y[(comp * 3 + 2) * 27 + t] = fma(SbT[2], gm[6], fma(SbT[1], gm[3], SbT[0] * gm[0]));
    }
    {
      const double *gd = X + 324 + a + 9 * c;
      Y[243 + t] = fma(PbT[2], gd[6], fma(PbT[1], gd[3], PbT[0] * gd[0]));
    }
    wave_fence();
    // ---- x: -> node (a, b, c)
    double rK[3], rM[3];
#pragma unroll
    for (int comp = 0; comp < 3; ++comp) {
      const double *h0 = Y + (comp * 3) * 27 + 3 * b + 9 * c, *h1 = h0 + 27, *hm = h0 + 54;
      rK[comp] = fma(DaT[2], h0[2], fma(DaT[1], h0[1], DaT[0] * h0[0])) + fma(SaT[2], h1[2], fma(SaT[1], h1[1], SaT[0] * h1[0]));
      rM[comp] = fma(SaT[2], hm[2], fma(SaT[1], hm[1], SaT[0] * hm[0]));
    }
    const double *hd = Y + 243 + 3 * b + 9 * c;
    const double rP = prm.pdg ? rPdg : fma(PaT[2], hd[2], fma(PaT[1], hd[1], PaT[0] * hd[0]));

    // ---- distribute_local_to_global: constrained velocity rows stay 0.  A DoF on a face shared with a
    // neighbouring cell is first touched by the cell whose colour bits are 0 in all shared directions.
    if constexpr (FUSED) {
#pragma unroll
      for (int o = 0; o < MAXSRC; ++o)
        if (o < prm.nout) {
          const double kU = prm.fKu[o][src], kM = prm.fM[o][src];
#pragma unroll
          for (int comp = 0; comp < 3; ++comp) accU[o][comp] = fma(kU, rK[comp], fma(kM, rM[comp], accU[o][comp]));
          accP[o] = fma(prm.fKp[o][src], rP, accP[o]);
        }
    }
    if (active && src == nsrc - 1) {
      const bool fu = !((a == 0 && cx > 0 && px) || (a == 2 && cx < prm.ncx - 1 && px) ||
                        (b == 0 && cy > 0 && py) || (b == 2 && cy < prm.ncy - 1 && py) ||
                        (c == 0 && cz > 0 && pz) || (c == 2 && cz < prm.ncz - 1 && pz));
      const bool fp = prm.pdg || !((a == 0 && cx > 0 && px) || (a == 1 && cx < prm.ncx - 1 && px) ||
                                    (b == 0 && cy > 0 && py) || (b == 1 && cy < prm.ncy - 1 && py) ||
                                    (c == 0 && cz > 0 && pz) || (c == 1 && cz < prm.ncz - 1 && pz));
      if constexpr (FUSED) {
#pragma unroll
        for (int o = 0; o < MAXSRC; ++o)
          if (o < prm.nout) {
            if (prm.out_u[o]) {
              double *d = prm.out_u[o] + gu;
              if (prm.store_u[o] && fu) {
#pragma unroll
                for (int comp = 0; comp < 3; ++comp) d[comp * prm.Nu] = con ? 0.0 : accU[o][comp];
              } else if (!con) {
                double v[3];
#pragma unroll
                for (int comp = 0; comp < 3; ++comp) v[comp] = d[comp * prm.Nu];
#pragma unroll
                for (int comp = 0; comp < 3; ++comp) d[comp * prm.Nu] = v[comp] + accU[o][comp];
              }
            }
            if (pnode && prm.out_p[o]) {
              double *d = prm.out_p[o] + gp;
              if (prm.store_p[o] && fp) *d = accP[o];
              else *d += accP[o];
            }
          }
      } else
      for (int j = 0; j < prm.nout; ++j) {
        if (prm.out_u[j]) {
          double *d = prm.out_u[j] + gu;
          if (prm.store_u[j] && fu) {
#pragma unroll
            for (int comp = 0; comp < 3; ++comp)
              d[comp * prm.Nu] = con ? 0.0 : prm.wKu[j] * rK[comp] + prm.wM[j] * rM[comp];
          } else if (!con) {
            double v[3];
#pragma unroll
            for (int comp = 0; comp < 3; ++comp) v[comp] = d[comp * prm.Nu];
#pragma unroll
            for (int comp = 0; comp < 3; ++comp) d[comp * prm.Nu] = v[comp] + (prm.wKu[j] * rK[comp] + prm.wM[j] * rM[comp]);
          }
        }
        if (pnode && prm.out_p[j]) {
          double *d = prm.out_p[j] + gp;
          if (prm.store_p[j] && fp) *d = prm.wKp[j] * rP;
          else *d += prm.wKp[j] * rP;
        }
      }
    }
    if (FUSED && src == nsrc - 1) {
#pragma unroll
      for (int o = 0; o < (FUSED ? MAXSRC : 1); ++o) accU[o][0] = accU[o][1] = accU[o][2] = accP[o] = 0.0;
    }
    wave_fence(); // the next cell's gather overwrites X
  }
}


// ---- boundary faces of the linear operator (LoopType::Full, reference include/operators.h:1640-1741) ----
// Weak (Nitsche) faces: v <- -nu grad u n + p n + gamma1/h u + gamma2/h n (u.n), dv/dn <- -nu u, q <- -u.n with
// gamma1 = nu penalty1, gamma2 = penalty2 (1220-1221) and h = sqrt(face area) (get_h_face, 184-209); outflow faces add
// nothing to the linear operator (1680-1711: the back-flow term carries a factor 0.0, the rest is nonlinear-only).
// The same kernel evaluates StokesNitscheMatrixFreeOperator::vmult (1898-1940): the functional of the Dirichlet data g.
// One half-wave per boundary CELL (a cell on several weak faces is handled once, by its lowest face, for all of them);
// eight colour launches after the cell loop's, plain read-add-write: no atomics, reproducible.
struct BoundaryParams {
  int weak_mask;
  double gamma1, gamma2;
  int foff[7];             // first work item (cell of a face, t1 fastest) of every face, [6] = total
  const double *g;         // rhs mode: Dirichlet data at the face quadrature points [point][3]; nullptr: operator mode
  double Eu[6], EDu[6], Ep[4]; // end-point tables [s * n + a]: FE_Q(2) values / derivatives, FE_Q(1) values at 0 and 1
};

template <bool FUSED>
__global__ __launch_bounds__(256) void stokes_boundary_kernel(const StokesParams prm, const BoundaryParams bp)
{
  __shared__ double tS[9], tD[9], tP[6], tE[6], tED[6], tEP[4], tX[3], tW[3];
  __shared__ double sX[8][89], sF[8][9][7], sG[8][9][12], sJ[8][9];
  if (threadIdx.x < 9) { tS[threadIdx.x] = prm.Su[threadIdx.x]; tD[threadIdx.x] = prm.Du[threadIdx.x]; }
  if (threadIdx.x < 6) { tP[threadIdx.x] = prm.Sp[threadIdx.x]; tE[threadIdx.x] = bp.Eu[threadIdx.x]; tED[threadIdx.x] = bp.EDu[threadIdx.x]; }
  if (threadIdx.x < 4) tEP[threadIdx.x] = bp.Ep[threadIdx.x];
  if (threadIdx.x < 3) { tX[threadIdx.x] = prm.xq[threadIdx.x]; tW[threadIdx.x] = prm.wq[threadIdx.x]; }
  __syncthreads();
  const int slot = threadIdx.x >> 5, t32 = threadIdx.x & 31;
  const bool lane27 = t32 < 27;
  const int t = lane27 ? t32 : 0;
  const int a = t % 3, b = (t / 3) % 3, c = t / 9;
  const bool pnode = prm.pdg ? t32 < 4 : (lane27 && a < 2 && b < 2 && c < 2);
  const int pslot = prm.pdg ? t32 : a + 2 * b + 4 * c;
  const int nc[3] = {prm.ncx, prm.ncy, prm.ncz};
  double *X = sX[slot];
  for (long long item = (long long)blockIdx.x * 8 + slot; item - slot < bp.foff[6]; item += (long long)gridDim.x * 8) {
    // (all half-waves of the workgroup run the same number of rounds: nothing below is a workgroup barrier, but keep it uniform)
    bool ok = item < bp.foff[6];
    int f0 = 0;
    for (int f = 0; f < 6; ++f)
      if (ok && item >= bp.foff[f] && item < bp.foff[f + 1]) f0 = f;
    int cc[3] = {0, 0, 0};
    {
      const int d = f0 >> 1, s = f0 & 1, t1 = d == 0 ? 1 : 0, t2 = d == 2 ? 1 : 2;
      const long long e = ok ? item - bp.foff[f0] : 0;
      cc[d] = s ? nc[d] - 1 : 0;
      cc[t1] = int(e % nc[t1]);
      cc[t2] = int(e / nc[t1]);
    }
    const int cx = cc[0], cy = cc[1], cz = cc[2];
    ok = ok && ((cx & 1) + 2 * (cy & 1) + 4 * (cz & 1)) == prm.colour;
    // the cell's weak faces; it is handled by the lowest of them
    int faces = 0;
    for (int f = 0; f < 6; ++f) {
      const int d = f >> 1, s = f & 1;
      if ((bp.weak_mask >> f & 1) && cc[d] == (s ? nc[d] - 1 : 0)) faces |= 1 << f;
    }
    ok = ok && (faces & ((1 << f0) - 1)) == 0;
    if (!__builtin_amdgcn_readfirstlane(__ballot(ok) != 0)) continue; // (wave-uniform skip only: the two half-waves fence together)
    const int ix = 2 * cx + a, iy = 2 * cy + b, iz = 2 * cz + c;
    const bool con = constrained_u(prm, ix, iy, iz);
    const long long gu = ix + (long long)prm.ndu[0] * (iy + (long long)prm.ndu[1] * iz);
    const long long gp = prm.pdg ? (cx + (long long)prm.ncx * (cy + (long long)prm.ncy * cz)) * 4 + (t32 & 3)
                                 : (cx + (a < 2 ? a : 1)) + (long long)prm.ndp[0] * ((cy + (b < 2 ? b : 1)) + (long long)prm.ndp[1] * (cz + (c < 2 ? c : 1)));
    double accU[FUSED ? MAXSRC : 1][3], accP[FUSED ? MAXSRC : 1];
#pragma unroll
    for (int o = 0; o < (FUSED ? MAXSRC : 1); ++o) accU[o][0] = accU[o][1] = accU[o][2] = accP[o] = 0.0;
    const int nsrc = bp.g ? 1 : (FUSED ? prm.nsrc : 1);
    for (int src = 0; src < nsrc; ++src) {
      if (!bp.g) { // gather (read_dof_values: constrained velocity entries read as 0)
        const double *us = FUSED ? prm.us[src] : prm.u, *ps = FUSED ? prm.ps[src] : prm.p;
        if (lane27)
          for (int comp = 0; comp < 3; ++comp) X[comp * 27 + t] = (ok && !con) ? us[comp * prm.Nu + gu] : 0.0;
        if (pnode) X[81 + pslot] = (ok && ps) ? ps[gp] : 0.0;
      }
      double rU[3] = {0, 0, 0}, rP = 0.0;
      for (int f = 0; f < 6; ++f) {
        if (!__builtin_amdgcn_readfirstlane(__ballot(ok && (faces >> f & 1)) != 0)) continue;
        const bool on = ok && (faces >> f & 1); // per half-wave
        const int d = f >> 1, s = f & 1, t1 = d == 0 ? 1 : 0, t2 = d == 2 ? 1 : 2;
        const int q1 = t32 % 3, q2 = (t32 / 3) % 3;
        // 1D tables of face point (q1, q2) / of any point q: value and derivative of node n along direction dir
        auto tv = [&](int dir, int qa, int qb, int n) { return dir == d ? tE[s * 3 + n] : tS[(dir == t1 ? qa : qb) * 3 + n]; };
        auto td = [&](int dir, int qa, int qb, int n) { return dir == d ? tED[s * 3 + n] : tD[(dir == t1 ? qa : qb) * 3 + n]; };
        auto tp = [&](int dir, int qa, int qb, int n) { return dir == d ? tEP[s * 2 + n] : tP[(dir == t1 ? qa : qb) * 2 + n]; };
        // FE_DGP(1) function j at face point (qa, qb): 1, l(xi), l(eta), l(zeta), l(x) = sqrt 3 (2 x - 1)
        auto dg = [&](int j, int qa, int qb) {
          if (j == 0) return 1.0;
          const int dir = j - 1;
          const double x = dir == d ? double(s) : tX[dir == t1 ? qa : qb];
          return 1.7320508075688772 * (2.0 * x - 1.0);
        };
        double Ji[3][3], nrm[3], JxW = 0.0;
        if (on && t32 < 9) { // geometry of this lane's face point
          double xi[3];
          xi[d] = s; xi[t1] = tX[q1]; xi[t2] = tX[q2];
          const double fx[2] = {1 - xi[0], xi[0]}, fy[2] = {1 - xi[1], xi[1]}, fz[2] = {1 - xi[2], xi[2]}, dd[2] = {-1.0, 1.0};
          double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
          const long long nvx = prm.ncx + 1, nvy = prm.ncy + 1;
          for (int k = 0; k < 2; ++k)
            for (int j = 0; j < 2; ++j)
              for (int i = 0; i < 2; ++i) {
                const double *V = prm.vertices + 3 * ((cx + i) + nvx * ((cy + j) + nvy * (long long)(cz + k)));
                for (int e = 0; e < 3; ++e) {
                  const double Ve = V[e];
                  J[e][0] += Ve * dd[i] * fy[j] * fz[k];
                  J[e][1] += Ve * fx[i] * dd[j] * fz[k];
                  J[e][2] += Ve * fx[i] * fy[j] * dd[k];
                }
              }
          const double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
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
          double m[3], len = 0.0;
          for (int k = 0; k < 3; ++k) {
            m[k] = (s ? 1.0 : -1.0) * (d == 0 ? Ji[0][k] : (d == 1 ? Ji[1][k] : Ji[2][k]));
            len += m[k] * m[k];
          }
          len = sqrt(len);
          for (int k = 0; k < 3; ++k) nrm[k] = m[k] / len;
          JxW = fabs(det) * len * tW[q1] * tW[q2];
          sJ[slot][t32] = JxW;
        }
        wave_fence();
        if (on && t32 < 9) {
          double area = 0.0;
          for (int q = 0; q < 9; ++q) area += sJ[slot][q];
          const double h = sqrt(area); // get_h_face: area^(1 / (dim - 1))
          double val[3], nd[3], pq;
          if (bp.g) { // operators.h:1921-1932
            const int c1 = cc[t1], c2 = cc[t2];
            long long pt = 0;
            for (int ff = 0; ff < f; ++ff)
              if (bp.weak_mask >> ff & 1) pt += 9ll * (bp.foff[ff + 1] - bp.foff[ff]);
            pt += 9ll * (c1 + (long long)nc[t1] * c2) + t32;
            const double g0 = bp.g[3 * pt], g1 = bp.g[3 * pt + 1], g2 = bp.g[3 * pt + 2];
            const double gq[3] = {g0, g1, g2};
            const double gn = g0 * nrm[0] + g1 * nrm[1] + g2 * nrm[2];
            for (int comp = 0; comp < 3; ++comp) {
              val[comp] = ((bp.gamma1 / h) * gq[comp] + (bp.gamma2 / h) * nrm[comp] * gn) * JxW;
              nd[comp] = -prm.nu * gq[comp] * JxW;
            }
            pq = -gn * JxW;
          } else { // operators.h:1720-1739
            double uval[3] = {0, 0, 0}, gref[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, pval = 0.0;
            for (int kc = 0; kc < 3; ++kc)
              for (int kb = 0; kb < 3; ++kb)
                for (int ka = 0; ka < 3; ++ka) {
                  const double sx = tv(0, q1, q2, ka), sy = tv(1, q1, q2, kb), sz = tv(2, q1, q2, kc);
                  const double dx = td(0, q1, q2, ka) * sy * sz, dy = sx * td(1, q1, q2, kb) * sz, dz = sx * sy * td(2, q1, q2, kc);
                  for (int comp = 0; comp < 3; ++comp) {
                    const double w = X[comp * 27 + ka + 3 * kb + 9 * kc];
                    gref[comp][0] += w * dx; gref[comp][1] += w * dy; gref[comp][2] += w * dz;
                    uval[comp] += w * sx * sy * sz;
                  }
                }
            if (prm.pdg) {
              for (int j = 0; j < 4; ++j) pval += X[81 + j] * dg(j, q1, q2);
            } else
            for (int kc = 0; kc < 2; ++kc)
              for (int kb = 0; kb < 2; ++kb)
                for (int ka = 0; ka < 2; ++ka)
                  pval += X[81 + ka + 2 * kb + 4 * kc] * tp(0, q1, q2, ka) * tp(1, q1, q2, kb) * tp(2, q1, q2, kc);
            double un = 0.0, gn[3];
            for (int comp = 0; comp < 3; ++comp) {
              gn[comp] = 0.0;
              for (int k = 0; k < 3; ++k)
                gn[comp] += (gref[comp][0] * Ji[0][k] + gref[comp][1] * Ji[1][k] + gref[comp][2] * Ji[2][k]) * nrm[k];
              un += uval[comp] * nrm[comp];
            }
            for (int comp = 0; comp < 3; ++comp) {
              val[comp] = (-prm.nu * gn[comp] + pval * nrm[comp] + (bp.gamma1 / h) * uval[comp] + (bp.gamma2 / h) * nrm[comp] * un) * JxW;
              nd[comp] = -prm.nu * uval[comp] * JxW;
            }
            pq = -un * JxW;
          }
          double *F = sF[slot][t32], *G = sG[slot][t32];
          for (int comp = 0; comp < 3; ++comp) { F[comp] = val[comp]; F[3 + comp] = nd[comp]; }
          F[6] = pq;
          for (int e = 0; e < 3; ++e)
            for (int k = 0; k < 3; ++k) G[3 * e + k] = Ji[e][k];
          for (int k = 0; k < 3; ++k) G[9 + k] = nrm[k];
        }
        wave_fence();
        if (on && lane27) { // integrate: test values and test normal derivatives of node (a, b, c)
          for (int q = 0; q < 9; ++q) {
            const int qa = q % 3, qb = q / 3;
            const double *F = sF[slot][q], *G = sG[slot][q];
            const double sx = tv(0, qa, qb, a), sy = tv(1, qa, qb, b), sz = tv(2, qa, qb, c);
            const double gr[3] = {td(0, qa, qb, a) * sy * sz, sx * td(1, qa, qb, b) * sz, sx * sy * td(2, qa, qb, c)};
            double dn = 0.0;
            for (int k = 0; k < 3; ++k) dn += (gr[0] * G[k] + gr[1] * G[3 + k] + gr[2] * G[6 + k]) * G[9 + k];
            const double v = sx * sy * sz;
            for (int comp = 0; comp < 3; ++comp) rU[comp] += v * F[comp] + dn * F[3 + comp];
            if (pnode) rP += (prm.pdg ? dg(t32 & 3, qa, qb) : tp(0, qa, qb, a) * tp(1, qa, qb, b) * tp(2, qa, qb, c)) * F[6];
          }
        }
        wave_fence(); // the next face reuses the point buffers
      }
      if (FUSED && !bp.g) {
#pragma unroll
        for (int o = 0; o < MAXSRC; ++o)
          if (o < prm.nout) {
            for (int comp = 0; comp < 3; ++comp) accU[o][comp] = fma(prm.fKu[o][src], rU[comp], accU[o][comp]);
            accP[o] = fma(prm.fKp[o][src], rP, accP[o]);
          }
      } else {
        for (int comp = 0; comp < 3; ++comp) accU[0][comp] = rU[comp];
        accP[0] = rP;
      }
      wave_fence(); // the next source overwrites X
    }
    // distribute_local_to_global (add): constrained velocity rows are not written
    if (ok && lane27) {
      for (int o = 0; o < prm.nout; ++o) {
        const double kU = (FUSED || bp.g) ? 1.0 : prm.wKu[o], kP = (FUSED || bp.g) ? 1.0 : prm.wKp[o];
        const int oa = (FUSED && !bp.g) ? o : 0;
        if (prm.out_u[o] && !con && kU != 0.0) {
          double *dptr = prm.out_u[o] + gu;
          for (int comp = 0; comp < 3; ++comp) dptr[comp * prm.Nu] += kU * accU[oa][comp];
        }
        if (pnode && prm.out_p[o] && kP != 0.0) prm.out_p[o][gp] += kP * accP[oa];
      }
    }
  }
}

} // namespace

struct stfem_stokes_ctx {
  int device = 0;
  int nc[3] = {0, 0, 0};
  int ndu[3] = {0, 0, 0}, ndp[3] = {0, 0, 0};
  long long Nu = 0, Np = 0;
  int dmask = 0;
  double nu = 1.0;
  double *d_vertices = nullptr;
  int n_cu = 256;
  StokesParams base;
  // weak (Nitsche) / outflow boundary faces (operators.h:1206-1211): bit f = 2 d + s
  int pspace = 0; // 0 = FE_Q(1), 1 = FE_DGP(1)
  int weak_mask = 0, outflow_mask = 0;
  double penalty1 = 20.0, penalty2 = 10.0;
  BoundaryParams bnd;
  double *d_g = nullptr; // Dirichlet data at the face quadrature points (stfem_stokes_nitsche_rhs)
  size_t g_points = 0;
  std::vector<double> h_vertices;
};

static thread_local char g_stokes_err[256] = "";
#define STOKES_TRY(call)                                                   \
  do {                                                                     \
    hipError_t e_ = (call);                                                \
    if (e_ != hipSuccess) {                                                \
      snprintf(g_stokes_err, sizeof(g_stokes_err), "%s: %s", #call, hipGetErrorString(e_)); \
      return STFEM_ERR_HIP;                                                \
    }                                                                      \
  } while (0)

extern "C" {

const char *stfem_stokes_last_hip_error(void) { return g_stokes_err; }

int stfem_stokes_create(const stfem_mesh_desc *mesh, int velocity_degree, double viscosity, stfem_stokes_ctx **out)
{
  return stfem_stokes_create_ex(mesh, velocity_degree, 0, viscosity, out);
}

int stfem_stokes_create_ex(const stfem_mesh_desc *mesh, int velocity_degree, int pressure_space, double viscosity, stfem_stokes_ctx **out)
{
  if (!mesh || !out || pressure_space < 0 || pressure_space > 1) return STFEM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (velocity_degree != 2) return STFEM_ERR_UNSUPPORTED; // Q2/Q1 (BASELINE configs[4]) only
  for (int d = 0; d < 3; ++d)
    if (mesh->ncell[d] < 1) return STFEM_ERR_INVALID_ARGUMENT;
  int ndev = 0;
  {
    const hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
      snprintf(g_stokes_err, sizeof(g_stokes_err), "hipGetDeviceCount: %s (%d devices)", hipGetErrorString(e), ndev);
      return STFEM_ERR_NO_DEVICE;
    }
  }
  if (mesh->device < 0 || mesh->device >= ndev) return STFEM_ERR_INVALID_ARGUMENT;
  STOKES_TRY(hipSetDevice(mesh->device));
  stfem_stokes_ctx *c = new (std::nothrow) stfem_stokes_ctx;
  if (!c) return STFEM_ERR_OUT_OF_MEMORY;
  c->device = mesh->device;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, mesh->device) == hipSuccess && prop.multiProcessorCount > 0) c->n_cu = prop.multiProcessorCount;
  }
  c->dmask = mesh->dirichlet_mask;
  c->nu = viscosity;
  for (int d = 0; d < 3; ++d) {
    c->nc[d] = mesh->ncell[d];
    c->ndu[d] = 2 * mesh->ncell[d] + 1;
    c->ndp[d] = mesh->ncell[d] + 1;
  }
  c->Nu = (long long)c->ndu[0] * c->ndu[1] * c->ndu[2];
  c->Np = (long long)c->ndp[0] * c->ndp[1] * c->ndp[2];
  c->pspace = pressure_space;
  if (pressure_space == 1) c->Np = 4ll * c->nc[0] * c->nc[1] * c->nc[2]; // FE_DGP(1): 1, x, y, z per cell
  const size_t nv = size_t(c->nc[0] + 1) * (c->nc[1] + 1) * (c->nc[2] + 1);
  std::vector<double> v(nv * 3);
  if (mesh->vertices) {
    std::memcpy(v.data(), mesh->vertices, nv * 3 * sizeof(double));
  } else {
    size_t o = 0;
    for (int k = 0; k <= c->nc[2]; ++k)
      for (int j = 0; j <= c->nc[1]; ++j)
        for (int i = 0; i <= c->nc[0]; ++i, ++o) {
          v[3 * o] = mesh->lower[0] + (mesh->upper[0] - mesh->lower[0]) * i / c->nc[0];
          v[3 * o + 1] = mesh->lower[1] + (mesh->upper[1] - mesh->lower[1]) * j / c->nc[1];
          v[3 * o + 2] = mesh->lower[2] + (mesh->upper[2] - mesh->lower[2]) * k / c->nc[2];
        }
  }
  if (hipMalloc(&c->d_vertices, nv * 3 * sizeof(double)) != hipSuccess) {
    delete c;
    return STFEM_ERR_OUT_OF_MEMORY;
  }
  if (hipMemcpy(c->d_vertices, v.data(), nv * 3 * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(c->d_vertices);
    delete c;
    return STFEM_ERR_HIP;
  }
  c->h_vertices = v;
  std::memset(&c->bnd, 0, sizeof(c->bnd));
  // 1D tables: FE_Q(2) and FE_Q(1) on Gauss-Lobatto nodes at the 3 Gauss points
  StokesParams &b = c->base;
  std::memset(&b, 0, sizeof(b));
  const stfem::ShapeTables tu = stfem::make_shape_tables(2), tp = stfem::make_shape_tables(1);
  std::vector<double> xq, wq;
  stfem::gauss_rule(3, xq, wq);
  stfem::Mat Sp, Gp;
  stfem::lagrange_tables(tp.nodes, xq, Sp, Gp);
  for (int i = 0; i < 9; ++i) { b.Su[i] = tu.S[i]; b.Du[i] = tu.D[i]; }
  for (int i = 0; i < 6; ++i) b.Sp[i] = Sp[i];
  for (int i = 0; i < 3; ++i) { b.xq[i] = xq[i]; b.wq[i] = wq[i]; }
  b.vertices = c->d_vertices;
  b.ncx = c->nc[0]; b.ncy = c->nc[1]; b.ncz = c->nc[2];
  for (int d = 0; d < 3; ++d) { b.ndu[d] = c->ndu[d]; b.ndp[d] = c->ndp[d]; }
  b.Nu = c->Nu; b.Np = c->Np;
  b.dmask = c->dmask;
  b.nu = c->nu;
  b.cart = mesh->vertices ? 0 : 1;
  b.pdg = c->pspace;
  for (int i = 0; i < 3; ++i) b.l1q[i] = std::sqrt(3.0) * (2.0 * xq[i] - 1.0);
  b.interleave = 1;
  if (const char *e = getenv("STFEM_STOKES_INTERLEAVE")) b.interleave = atoi(e) != 0;
  b.detJ = 1.0;
  for (int d = 0; d < 3; ++d) {
    const double h = (mesh->upper[d] - mesh->lower[d]) / c->nc[d];
    b.hinv[d] = 1.0 / h;
    b.detJ *= h;
  }
  *out = c;
  return STFEM_OK;
}

void stfem_stokes_destroy(stfem_stokes_ctx *c)
{
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->d_vertices) (void)hipFree(c->d_vertices);
  if (c->d_g) (void)hipFree(c->d_g);
  delete c;
}

int64_t stfem_stokes_n_velocity_dofs(const stfem_stokes_ctx *c) { return c ? c->Nu : 0; }
int64_t stfem_stokes_n_pressure_dofs(const stfem_stokes_ctx *c) { return c ? c->Np : 0; }

static size_t stokes_len(const stfem_stokes_ctx *c, int variable) { return variable == 0 ? size_t(3 * c->Nu) : size_t(c->Np); }

int stfem_stokes_vector_create(stfem_stokes_ctx *c, int variable, double **device_out)
{
  if (!c || !device_out || variable < 0 || variable > 1) return STFEM_ERR_INVALID_ARGUMENT;
  *device_out = nullptr;
  STOKES_TRY(hipSetDevice(c->device));
  double *d = nullptr;
  if (hipMalloc(&d, stokes_len(c, variable) * sizeof(double)) != hipSuccess) return STFEM_ERR_OUT_OF_MEMORY;
  if (hipMemset(d, 0, stokes_len(c, variable) * sizeof(double)) != hipSuccess) {
    (void)hipFree(d);
    return STFEM_ERR_HIP;
  }
  *device_out = d;
  return STFEM_OK;
}

void stfem_stokes_vector_destroy(stfem_stokes_ctx *c, double *device_vec)
{
  if (!c || !device_vec) return;
  (void)hipSetDevice(c->device);
  (void)hipFree(device_vec);
}

int stfem_stokes_vector_upload(stfem_stokes_ctx *c, int variable, double *device_vec, const double *host)
{
  if (!c || !device_vec || !host || variable < 0 || variable > 1) return STFEM_ERR_INVALID_ARGUMENT;
  STOKES_TRY(hipSetDevice(c->device));
  STOKES_TRY(hipMemcpy(device_vec, host, stokes_len(c, variable) * sizeof(double), hipMemcpyHostToDevice));
  return STFEM_OK;
}

int stfem_stokes_vector_download(stfem_stokes_ctx *c, int variable, const double *device_vec, double *host)
{
  if (!c || !device_vec || !host || variable < 0 || variable > 1) return STFEM_ERR_INVALID_ARGUMENT;
  STOKES_TRY(hipSetDevice(c->device));
  STOKES_TRY(hipMemcpy(host, device_vec, stokes_len(c, variable) * sizeof(double), hipMemcpyDeviceToHost));
  return STFEM_OK;
}

static int stokes_boundary_launch(stfem_stokes_ctx *c, StokesParams &prm, const double *d_g, hipStream_t st)
{
  BoundaryParams bp = c->bnd;
  bp.g = d_g;
  const long long items = bp.foff[6];
  if (items == 0) return STFEM_OK;
  const unsigned grid = (unsigned)std::min<long long>((items + 7) / 8, 4ll * c->n_cu);
  (void)hipGetLastError();
  for (int colour = 0; colour < 8; ++colour) {
    prm.colour = colour;
    if (prm.nsrc > 1 && !d_g) hipLaunchKernelGGL(stokes_boundary_kernel<true>, dim3(grid), dim3(256), 0, st, prm, bp);
    else hipLaunchKernelGGL(stokes_boundary_kernel<false>, dim3(grid), dim3(256), 0, st, prm, bp);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_stokes_err, sizeof(g_stokes_err), "stokes_boundary_kernel: %s", hipGetErrorString(e));
    return STFEM_ERR_HIP;
  }
  return STFEM_OK;
}

static int stokes_launch(stfem_stokes_ctx *c, StokesParams &prm, hipStream_t st)
{
  (void)hipGetLastError();
  for (int colour = 0; colour < 8; ++colour) { // ascending: see store_u / store_p
    const long long n = (long long)((c->nc[0] - (colour & 1) + 1) / 2) * ((c->nc[1] - ((colour >> 1) & 1) + 1) / 2) *
                        ((c->nc[2] - (colour >> 2) + 1) / 2);
    if (n == 0) continue;
    prm.colour = colour;
    // persistent workgroups: exactly as many as stay resident (measured on 64^3 cells, cG(1): 2 per CU 0.41 ms, 3: 0.50, 4: 0.43,
    // 8: 0.45, one workgroup per 8 cells: 0.52 - long runs keep the prefetch of the next cell's DoFs going and leave no partial round)
    static const int grid_env = [] {
      const char *e = getenv("STFEM_STOKES_GRID"); // workgroups per CU of a colour launch (experiments)
      return e ? std::max(1, atoi(e)) : 0;
    }();
    const void *kern = prm.nsrc > 1 ? (prm.cart ? (const void *)stokes_cell_kernel<true, true> : (const void *)stokes_cell_kernel<false, true>)
                                    : (prm.cart ? (const void *)stokes_cell_kernel<true, false> : (const void *)stokes_cell_kernel<false, false>);
    // resident workgroups per CU of the four instantiations, asked once (not on the launch path)
    static int resident_of[4] = {0, 0, 0, 0};
    int &resident = resident_of[(prm.nsrc > 1 ? 2 : 0) + (prm.cart ? 1 : 0)];
    if (resident < 1 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, kern, 256, 0) != hipSuccess || resident < 1)) resident = 2;
    const unsigned grid = (unsigned)std::min<long long>((n + 7) / 8, (long long)c->n_cu * (grid_env ? grid_env : resident));
    void *args[] = {(void *)&prm};
    (void)hipLaunchKernel(kern, dim3(grid), dim3(256), args, 0, st);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_stokes_err, sizeof(g_stokes_err), "stokes_cell_kernel: %s", hipGetErrorString(e));
    return STFEM_ERR_HIP;
  }
  // LoopType::Full: the boundary-face loop of the same vmult (the mass operator has none)
  bool k_part = false;
  if (prm.nsrc > 1) {
    for (int o = 0; o < prm.nout; ++o)
      for (int q = 0; q < prm.nsrc; ++q) k_part = k_part || prm.fKu[o][q] != 0.0 || prm.fKp[o][q] != 0.0;
  } else
    for (int o = 0; o < prm.nout; ++o) k_part = k_part || prm.wKu[o] != 0.0 || prm.wKp[o] != 0.0;
  if (c->weak_mask && k_part) return stokes_boundary_launch(c, prm, nullptr, st);
  return STFEM_OK;
}

int stfem_stokes_vmult(stfem_stokes_ctx *c, double *dst_u, double *dst_p, const double *src_u,
                       const double *src_p, void *stream)
{
  if (!c || !dst_u || !dst_p || !src_u || !src_p) return STFEM_ERR_INVALID_ARGUMENT;
  if (dst_u == src_u || dst_p == src_p) return STFEM_ERR_ALIAS;
  STOKES_TRY(hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  StokesParams prm = c->base;
  prm.u = src_u; prm.p = src_p;
  prm.nout = 1;
  prm.out_u[0] = dst_u; prm.out_p[0] = dst_p;
  prm.store_u[0] = prm.store_p[0] = 1; // dst is overwritten
  prm.wKu[0] = 1.0; prm.wKp[0] = 1.0; prm.wM[0] = 0.0;
  return stokes_launch(c, prm, st);
}

int stfem_stokes_mass_vmult(stfem_stokes_ctx *c, double *dst_u, const double *src_u, void *stream)
{
  if (!c || !dst_u || !src_u) return STFEM_ERR_INVALID_ARGUMENT;
  if (dst_u == src_u) return STFEM_ERR_ALIAS;
  STOKES_TRY(hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  StokesParams prm = c->base;
  prm.u = src_u; prm.p = nullptr;
  prm.nout = 1;
  prm.out_u[0] = dst_u; prm.out_p[0] = nullptr;
  prm.store_u[0] = 1;
  prm.wKu[0] = 0.0; prm.wKp[0] = 0.0; prm.wM[0] = 1.0;
  return stokes_launch(c, prm, st);
}

int stfem_stokes_st_vmult(stfem_stokes_ctx *c, int n_timesteps_at_once, int n_timedofs, int variable_major,
                          const double *Alpha, const double *Beta, double *const *dst_blocks,
                          const double *const *src_blocks, void *stream)
{
  if (!c || !Alpha || !Beta || !dst_blocks || !src_blocks || n_timesteps_at_once < 1 || n_timedofs < 1)
    return STFEM_ERR_INVALID_ARGUMENT;
  const int nt = n_timedofs, ns = n_timesteps_at_once, nb = 2 * nt * ns;
  auto index = [&](int it, int v, int d) { // BlockSlice::index, fe_time.h:956-967
    return variable_major ? it * (2 * nt) + v * nt + d : it * (2 * nt) + d * 2 + v;
  };
  for (int j = 0; j < nb; ++j) {
    if (!dst_blocks[j] || !src_blocks[j]) return STFEM_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < nb; ++i)
      if (dst_blocks[j] == src_blocks[i]) return STFEM_ERR_ALIAS;
  }
  STOKES_TRY(hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  // dst = 0.0 (operators.h:833): the first launch that reaches a block overwrites it; blocks no launch
  // reaches are zeroed at the end
  std::vector<char> written(nb, 0);
  const double eps10 = 10 * std::numeric_limits<double>::epsilon(); // internal::scatter, operators.h:106
  // up to MAXSRC time dofs: ONE set of colour launches - every cell is evaluated for all sources, the destinations are written once
  static const bool fused_ok = [] {
    const char *e = getenv("STFEM_STOKES_FUSED");
    return !e || atoi(e) != 0;
  }();
  if (fused_ok && ns * nt >= 2 && ns * nt <= MAXSRC) {
    StokesParams prm = c->base;
    prm.nsrc = ns * nt;
    prm.nout = ns * nt;
    for (int it = 0; it < ns; ++it)
      for (int id = 0; id < nt; ++id) {
        const int sidx = it * nt + id, i = index(it, 0, id);
        prm.us[sidx] = src_blocks[index(it, 0, id)];
        prm.ps[sidx] = src_blocks[index(it, 1, id)];
        for (int jt = 0; jt < ns; ++jt)
          for (int jd = 0; jd < nt; ++jd) {
            const int o = jt * nt + jd, ju = index(jt, 0, jd), jp = index(jt, 1, jd);
            const double aU = Alpha[size_t(ju) * nb + i], aP = Alpha[size_t(jp) * nb + i], bU = Beta[size_t(ju) * nb + i];
            prm.fKu[o][sidx] = std::abs(aU) > eps10 ? aU : 0.0; // entries below the threshold are skipped (operators.h:91-110)
            prm.fKp[o][sidx] = std::abs(aP) > eps10 ? aP : 0.0;
            prm.fM[o][sidx] = std::abs(bU) > eps10 ? bU : 0.0;
          }
      }
    for (int jt = 0; jt < ns; ++jt)
      for (int jd = 0; jd < nt; ++jd) {
        const int o = jt * nt + jd;
        prm.out_u[o] = dst_blocks[index(jt, 0, jd)];
        prm.out_p[o] = dst_blocks[index(jt, 1, jd)];
        prm.store_u[o] = prm.store_p[o] = 1; // dst = 0.0 + the sums: every destination is overwritten
      }
    return stokes_launch(c, prm, st);
  }
  for (int it = 0; it < ns; ++it)
    for (int id = 0; id < nt; ++id) {
      const int i = index(it, 0, id); // the velocity column drives all scatters (operators.h:851-862)
      StokesParams prm = c->base;
      prm.u = src_blocks[index(it, 0, id)];
      prm.p = src_blocks[index(it, 1, id)];
      prm.nout = 0;
      auto flush = [&]() -> int {
        if (prm.nout == 0) return STFEM_OK;
        const int rc = stokes_launch(c, prm, st);
        prm.nout = 0;
        return rc;
      };
      for (int jt = 0; jt < ns; ++jt)
        for (int jd = 0; jd < nt; ++jd) {
          const int ju = index(jt, 0, jd), jp = index(jt, 1, jd);
          const double aU = Alpha[size_t(ju) * nb + i], aP = Alpha[size_t(jp) * nb + i], bU = Beta[size_t(ju) * nb + i];
          const bool useU = std::abs(aU) > eps10, useP = std::abs(aP) > eps10, useM = std::abs(bU) > eps10;
          if (!useU && !useP && !useM) continue;
          const int o = prm.nout++;
          prm.out_u[o] = (useU || useM) ? dst_blocks[ju] : nullptr;
          prm.out_p[o] = useP ? dst_blocks[jp] : nullptr;
          prm.store_u[o] = prm.out_u[o] && !written[ju];
          prm.store_p[o] = prm.out_p[o] && !written[jp];
          if (prm.out_u[o]) written[ju] = 1;
          if (prm.out_p[o]) written[jp] = 1;
          prm.wKu[o] = useU ? aU : 0.0;
          prm.wKp[o] = useP ? aP : 0.0;
          prm.wM[o] = useM ? bU : 0.0;
          if (prm.nout == MAXOUT) {
            const int rc = flush();
            if (rc != STFEM_OK) return rc;
          }
        }
      const int rc = flush();
      if (rc != STFEM_OK) return rc;
    }
  for (int it = 0; it < ns; ++it)
    for (int d = 0; d < nt; ++d) {
      if (!written[index(it, 0, d)]) STOKES_TRY(hipMemsetAsync(dst_blocks[index(it, 0, d)], 0, sizeof(double) * 3 * c->Nu, st));
      if (!written[index(it, 1, d)]) STOKES_TRY(hipMemsetAsync(dst_blocks[index(it, 1, d)], 0, sizeof(double) * c->Np, st));
    }
  return STFEM_OK;
}

// SystemMatrixStokes::vmult_slice_add (operators.h:748-781): the n x 1 case used for the right-hand
// side: src = one (velocity, pressure) pair, dst[index(it,v,id)] += Gamma(index(it,v,id), 0) * (K_S src)_v
// and dst[index(it,0,id)] += Zeta(index(it,0,id), 0) * M u.  dst is NOT zeroed.
int stfem_stokes_st_vmult_slice_add(stfem_stokes_ctx *c, int n_timesteps_at_once, int n_timedofs, int variable_major,
                                    const double *Gamma, const double *Zeta, double *const *dst_blocks,
                                    const double *src_u, const double *src_p, void *stream)
{
  if (!c || !Gamma || !Zeta || !dst_blocks || !src_u || !src_p || n_timesteps_at_once < 1 || n_timedofs < 1)
    return STFEM_ERR_INVALID_ARGUMENT;
  const int nt = n_timedofs, ns = n_timesteps_at_once, nb = 2 * nt * ns;
  auto index = [&](int it, int v, int d) { return variable_major ? it * (2 * nt) + v * nt + d : it * (2 * nt) + d * 2 + v; };
  for (int j = 0; j < nb; ++j) {
    if (!dst_blocks[j]) return STFEM_ERR_INVALID_ARGUMENT;
    if (dst_blocks[j] == src_u || dst_blocks[j] == src_p) return STFEM_ERR_ALIAS;
  }
  STOKES_TRY(hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const double eps10 = 10 * std::numeric_limits<double>::epsilon();
  StokesParams prm = c->base;
  prm.u = src_u;
  prm.p = src_p;
  prm.nout = 0;
  for (int it = 0; it < ns; ++it)
    for (int id = 0; id < nt; ++id) {
      const int ju = index(it, 0, id), jp = index(it, 1, id);
      const double aU = Gamma[ju], aP = Gamma[jp], bU = Zeta[ju];
      const bool useU = std::abs(aU) > eps10, useP = std::abs(aP) > eps10, useM = std::abs(bU) > eps10;
      if (!useU && !useP && !useM) continue;
      const int o = prm.nout++;
      prm.out_u[o] = (useU || useM) ? dst_blocks[ju] : nullptr;
      prm.out_p[o] = useP ? dst_blocks[jp] : nullptr;
      prm.wKu[o] = useU ? aU : 0.0;
      prm.wKp[o] = useP ? aP : 0.0;
      prm.wM[o] = useM ? bU : 0.0;
      if (prm.nout == MAXOUT) {
        const int rc = stokes_launch(c, prm, st);
        if (rc != STFEM_OK) return rc;
        prm.nout = 0;
      }
    }
  return prm.nout ? stokes_launch(c, prm, st) : STFEM_OK;
}

// ---- weak boundary conditions (operators.h:1206-1211, 1220-1221; StokesNitscheMatrixFreeOperator 1768-1951) ----
int stfem_stokes_set_weak_boundaries(stfem_stokes_ctx *c, int weak_mask, int outflow_mask, double penalty1, double penalty2)
{
  if (!c || weak_mask < 0 || weak_mask > 63 || outflow_mask < 0 || outflow_mask > 63) return STFEM_ERR_INVALID_ARGUMENT;
  // a face in both sets takes the outflow branch in the reference (1680: checked first), i.e. no term in the linear operator
  c->outflow_mask = outflow_mask;
  c->weak_mask = weak_mask & ~outflow_mask;
  c->penalty1 = penalty1;
  c->penalty2 = penalty2;
  BoundaryParams &b = c->bnd;
  std::memset(&b, 0, sizeof(b));
  b.weak_mask = c->weak_mask;
  b.gamma1 = c->nu * penalty1;
  b.gamma2 = penalty2;
  int off = 0;
  for (int f = 0; f < 6; ++f) {
    b.foff[f] = off;
    const int d = f / 2, t1 = d == 0 ? 1 : 0, t2 = d == 2 ? 1 : 2;
    if (c->weak_mask >> f & 1) off += c->nc[t1] * c->nc[t2];
  }
  b.foff[6] = off;
  const stfem::ShapeTables tu = stfem::make_shape_tables(2), tp = stfem::make_shape_tables(1);
  const std::vector<double> ends = {0.0, 1.0};
  stfem::Mat Eu, EDu, Ep, EDp;
  stfem::lagrange_tables(tu.nodes, ends, Eu, EDu);
  stfem::lagrange_tables(tp.nodes, ends, Ep, EDp);
  for (int i = 0; i < 6; ++i) { b.Eu[i] = Eu[i]; b.EDu[i] = EDu[i]; }
  for (int i = 0; i < 4; ++i) b.Ep[i] = Ep[i];
  return STFEM_OK;
}

int64_t stfem_stokes_n_face_points(const stfem_stokes_ctx *c) { return c ? 9ll * c->bnd.foff[6] : 0; }

int stfem_stokes_face_points(const stfem_stokes_ctx *c, double *out)
{
  if (!c || !out) return STFEM_ERR_INVALID_ARGUMENT;
  std::vector<double> xq, wq;
  stfem::gauss_rule(3, xq, wq);
  const long long nvx = c->nc[0] + 1, nvy = c->nc[1] + 1;
  size_t pt = 0;
  for (int f = 0; f < 6; ++f) {
    if (!(c->weak_mask >> f & 1)) continue;
    const int d = f / 2, s = f % 2, t1 = d == 0 ? 1 : 0, t2 = d == 2 ? 1 : 2;
    for (int c2 = 0; c2 < c->nc[t2]; ++c2)
      for (int c1 = 0; c1 < c->nc[t1]; ++c1) {
        int cc[3];
        cc[d] = s ? c->nc[d] - 1 : 0; cc[t1] = c1; cc[t2] = c2;
        for (int q2 = 0; q2 < 3; ++q2)
          for (int q1 = 0; q1 < 3; ++q1, ++pt) {
            double xi[3];
            xi[d] = s; xi[t1] = xq[q1]; xi[t2] = xq[q2];
            double x[3] = {0, 0, 0};
            for (int k = 0; k < 2; ++k)
              for (int j = 0; j < 2; ++j)
                for (int i = 0; i < 2; ++i) {
                  const double w = (i ? xi[0] : 1 - xi[0]) * (j ? xi[1] : 1 - xi[1]) * (k ? xi[2] : 1 - xi[2]);
                  const double *V = c->h_vertices.data() + 3 * ((cc[0] + i) + nvx * ((cc[1] + j) + nvy * (long long)(cc[2] + k)));
                  for (int e = 0; e < 3; ++e) x[e] += w * V[e];
                }
            for (int e = 0; e < 3; ++e) out[3 * pt + e] = x[e];
          }
      }
  }
  return STFEM_OK;
}

int stfem_stokes_nitsche_rhs(stfem_stokes_ctx *c, const double *g_at_face_points, double *dst_u, double *dst_p, void *stream)
{
  if (!c || !g_at_face_points || !dst_u || !dst_p) return STFEM_ERR_INVALID_ARGUMENT;
  const size_t npts = size_t(stfem_stokes_n_face_points(c));
  if (npts == 0) return STFEM_OK; // no Dirichlet functions: vmult does nothing (operators.h:1836)
  STOKES_TRY(hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (c->g_points < npts) {
    if (c->d_g) STOKES_TRY(hipFree(c->d_g));
    c->d_g = nullptr;
    c->g_points = 0;
    if (hipMalloc(&c->d_g, npts * 3 * sizeof(double)) != hipSuccess) return STFEM_ERR_OUT_OF_MEMORY;
    c->g_points = npts;
  }
  STOKES_TRY(hipMemcpyAsync(c->d_g, g_at_face_points, npts * 3 * sizeof(double), hipMemcpyHostToDevice, st));
  STOKES_TRY(hipStreamSynchronize(st)); // (the caller's host array may go away)
  StokesParams prm = c->base;
  prm.u = nullptr; prm.p = nullptr;
  prm.nsrc = 0;
  prm.nout = 1;
  prm.out_u[0] = dst_u; prm.out_p[0] = dst_p;
  prm.wKu[0] = prm.wKp[0] = 1.0;
  return stokes_boundary_launch(c, prm, c->d_g, st);
}

} // extern "C"
