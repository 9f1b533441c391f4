// Stokes two-field operator on MI355X (BASELINE configs[4]).
//
// Two implementations of the cell loop.  Axis-aligned uniform meshes: the Kronecker form further down ("axis-aligned uniform
// meshes") - the velocity components as the blocks of the scalar FE_Q(2) pencil sweep (csrc/stfem_pencil.hip), for one time dof
// with the pressure gradient term folded into that sweep, and the divergence as a marching gather kernel.  General meshes: the
// cell kernel described next.  On top of either: the weak (Nitsche) boundary faces (stokes_boundary_kernel) and the helpers of the
// pressure space the solver around the operator needs (end of the file).
//
// The cell kernel replaces, for the cell loop (LoopType::Cell: no weak boundary ids, delta0 = 0):
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
#include "stfem_internal.h"

#include <hip/hip_runtime.h>
#include <mutex>

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
// PDG: FE_DGP(1) pressure (a template parameter: the FE_Q(1) instantiations stay what they were)
template <bool CART, bool FUSED, bool PDG>
__global__ __launch_bounds__(256) void stokes_cell_kernel(const StokesParams prm)
{
  constexpr int RX = 351, RY = 351; // doubles per cell of the two regions (largest stage: 13 x 27)
  __shared__ double smem[8 * (RX + RY)];
  __shared__ double tS[9], tD[9], tP[6], tL[3]; // 1D tables [q*3+a], [q*3+a], [q*2+a]; l at the Gauss points
  if (threadIdx.x < 9) { tS[threadIdx.x] = prm.Su[threadIdx.x]; tD[threadIdx.x] = prm.Du[threadIdx.x]; }
  if (threadIdx.x < 6) tP[threadIdx.x] = prm.Sp[threadIdx.x];
  if (threadIdx.x < 3) tL[threadIdx.x] = prm.l1q[threadIdx.x];
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
  const bool pnode = PDG ? t32 < 4 : (lane27 && a < 2 && b < 2 && c < 2);
  const int pslot = PDG ? t32 : a + 2 * b + 4 * c; // its slot in the cell's pressure values X[81 ..]
  const double la = PDG ? tL[a] : 0.0, lb = PDG ? tL[b] : 0.0, lc = PDG ? tL[c] : 0.0; // DGP: the linear functions at this lane's quadrature point
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
    q.gp = PDG ? (q.cx + (long long)prm.ncx * (q.cy + (long long)prm.ncy * q.cz)) * 4 + (t32 & 3)
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
    if (PDG) {
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
    if (PDG) pval = pdgv[0] + la * pdgv[1] + lb * pdgv[2] + lc * pdgv[3];

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
    if (PDG && t32 < 4) { // (q_j, div u): the cell's own four test functions, summed over the 27 quadrature points
      const double *fd = Y + 9 * 27;
      for (int q = 0; q < 27; ++q) {
        const int qa = q % 3, qb = (q / 3) % 3, qc = q / 9;
        const double l = t32 == 0 ? 1.0 : tL[t32 == 1 ? qa : (t32 == 2 ? qb : qc)];
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
    const double rP = PDG ? rPdg : fma(PaT[2], hd[2], fma(PaT[1], hd[1], PaT[0] * hd[0]));

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
      const bool fp = PDG || !((a == 0 && cx > 0 && px) || (a == 1 && cx < prm.ncx - 1 && px) ||
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


// ---- axis-aligned uniform meshes: the operator in its Kronecker form (round 3) ----
// On boxes of identical cells the velocity part  nu K u_c + wM M u_c  of every component is the SCALAR space-time operator of
// FE_Q(2): it runs as the scalar pencil sweep (stfem_st_vmult on a Q2 context: owner-writes, every DoF stored once), the
// components being blocks.  What is left of the cell loop (operators.h:1547-1570) is the coupling
//     out_u_c -= B_c^T p,   out_p = sum_c B_c u_c,   B_c = (q, d u_c / d x_c),
// whose cell matrices are Kronecker products of 1D mixed matrices (N = int phi_a psi_j, C = int phi_a' psi_j), so both run in
// GATHER form - one thread per destination DoF sums what its <= 8 cells contribute, in a fixed order: no colours, no atomics,
// every destination touched once.  (Measured on 64^3 cells the eight colour launches of the cell kernel were bound by their
// access pattern and LDS traffic: profiles/r2/stokes.)
struct CouplingParams {
  int ncx, ncy, ncz;
  int ndu[3], ndp[3];
  long long Nu;
  int dmask, pdg;
  double h[3];
  // 1D reference integrals, Q2 node a: FE_Q(1): N[a][j], C[a][j], j = 0, 1; FE_DGP(1): N[a][0] = int phi_a, N[a][1] = int l phi_a (same for C)
  double N[3][2], C[3][2];
  int nsrc, nout;
  const double *u[MAXSRC], *p[MAXSRC];
  double *out_u[MAXOUT], *out_p[MAXOUT];
  double wKu[MAXOUT][MAXSRC], wKp[MAXOUT][MAXSRC]; // [output][source]
  int store_p[MAXOUT];
};

// FE_Q(1): the pressure nodes a velocity line node i couples to: first index p0, weights of up to three (value / derivative forms)
__device__ __forceinline__ void q1_row(const CouplingParams &P, int i, int nc, int &p0, double (&wn)[3], double (&wc)[3])
{
  if (i & 1) { // midpoint of cell c
    p0 = i >> 1;
    wn[0] = P.N[1][0]; wn[1] = P.N[1][1]; wn[2] = 0.0;
    wc[0] = P.C[1][0]; wc[1] = P.C[1][1]; wc[2] = 0.0;
  } else { // vertex between cells c - 1 (its node 2) and c (its node 0)
    const int c = i >> 1;
    const bool lo = c > 0, hi = c < nc;
    p0 = c - 1;
    wn[0] = lo ? P.N[2][0] : 0.0; wn[1] = (lo ? P.N[2][1] : 0.0) + (hi ? P.N[0][0] : 0.0); wn[2] = hi ? P.N[0][1] : 0.0;
    wc[0] = lo ? P.C[2][0] : 0.0; wc[1] = (lo ? P.C[2][1] : 0.0) + (hi ? P.C[0][0] : 0.0); wc[2] = hi ? P.C[0][1] : 0.0;
  }
}

// out_u[o][c][node] -= sum_s wKu[o][s] (B_c^T p_s)[node]: one thread per velocity node.  NS / NO: compile-time bounds of the
// source / destination loops (their accumulators then live in registers; with run-time bounds the generic instantiation
// needed 246 VGPRs and scratch)
template <int NS, int NO, bool PDG>
__global__ __launch_bounds__(256, NO <= 2 ? 3 : 2) void stokes_grad_kernel(const CouplingParams P)
{
  const long long node = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= P.Nu) return;
  const int ix = int(node % P.ndu[0]), iy = int((node / P.ndu[0]) % P.ndu[1]), iz = int(node / ((long long)P.ndu[0] * P.ndu[1]));
  const bool con = ((P.dmask & 1) && ix == 0) || ((P.dmask & 2) && ix == P.ndu[0] - 1) || ((P.dmask & 4) && iy == 0) ||
                   ((P.dmask & 8) && iy == P.ndu[1] - 1) || ((P.dmask & 16) && iz == 0) || ((P.dmask & 32) && iz == P.ndu[2] - 1);
  if (con) return; // constrained rows are not written (they hold the sweep's exact zero)
  double acc[NO][3];
#pragma unroll
  for (int o = 0; o < NO; ++o) acc[o][0] = acc[o][1] = acc[o][2] = 0.0;
  // the destination values this thread updates are fetched first: their latency runs beside the pressure gather's (the kernel is
  // bound by dependent memory round trips, not by bytes: profiles/r3/stokes)
  double old[NO <= 2 ? NO : 1][3];
  if constexpr (NO <= 2) {
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
      for (int c = 0; c < 3; ++c) old[o][c] = (o < P.nout && P.out_u[o]) ? P.out_u[o][c * P.Nu + node] : 0.0;
  }
  const int nc[3] = {P.ncx, P.ncy, P.ncz};
  const int idx[3] = {ix, iy, iz};
  if constexpr (!PDG) {
    int p0[3];
    double wn[3][3], wc[3][3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      q1_row(P, d == 0 ? ix : (d == 1 ? iy : iz), d == 0 ? P.ncx : (d == 1 ? P.ncy : P.ncz), p0[d], wn[d], wc[d]);
#pragma unroll
      for (int e = 0; e < 3; ++e) wn[d][e] *= P.h[d]; // the value forms carry the cell size, the derivative forms do not
    }
    // entries beyond the lattice carry weight 0: clamp their index and keep the loops free of branches (27 independent loads)
    int jx[3], jy[3], jz[3];
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      jx[e] = min(max(p0[0] + e, 0), P.ndp[0] - 1);
      jy[e] = min(max(p0[1] + e, 0), P.ndp[1] - 1);
      jz[e] = min(max(p0[2] + e, 0), P.ndp[2] - 1);
    }
    _Pragma("unroll 1") for (int s = 0; s < P.nsrc; ++s) { // (a run-time loop: only the destination loops need compile-time bounds)
      double g[3] = {0, 0, 0};
      const double *ps = P.p[s];
#pragma unroll
      for (int ez = 0; ez < 3; ++ez) { // separable sums (no table of the 81 weight products)
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
        for (int ey = 0; ey < 3; ++ey) {
          const double *row = ps + (long long)P.ndp[0] * (jy[ey] + (long long)P.ndp[1] * jz[ez]);
          double sc = 0.0, sn = 0.0;
#pragma unroll
          for (int ex = 0; ex < 3; ++ex) {
            const double pv = row[jx[ex]];
            sc = fma(wc[0][ex], pv, sc);
            sn = fma(wn[0][ex], pv, sn);
          }
          a0 = fma(wn[1][ey], sc, a0);
          a1 = fma(wc[1][ey], sn, a1);
          a2 = fma(wn[1][ey], sn, a2);
        }
        g[0] = fma(wn[2][ez], a0, g[0]);
        g[1] = fma(wn[2][ez], a1, g[1]);
        g[2] = fma(wc[2][ez], a2, g[2]);
        if constexpr (NO > 2) __builtin_amdgcn_sched_barrier(0); // (many destinations: nine loads in flight at a time keep the registers)
      }
      _Pragma("unroll") for (int o = 0; o < NO; ++o)
        if (o < P.nout)
          for (int c = 0; c < 3; ++c) acc[o][c] = fma(P.wKu[o][s], g[c], acc[o][c]);
    }
  } else {
    // FE_DGP(1), four functions per cell.  Per direction the node lies in up to two cells: slot 0 = the cell it is local node 1 (odd
    // index) or 2 (even index) of, slot 1 = the cell above an even node (local node 0).  A missing cell keeps a clamped index and
    // zero weights, so that the 8 x 4 coefficient loads are independent and the loops free of branches and of indexed reads of the
    // kernel arguments (round 3: the divergent loops over run-time slot counts took 55 us on 64^3 cells)
    int cell[3][2];
    double wN0[3][2], wN1[3][2], wC0[3][2], wC1[3][2];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int half = idx[d] >> 1;
      const bool odd = idx[d] & 1;
      const int below = odd ? half : half - 1;
      const bool v0 = below >= 0, v1 = !odd && half < nc[d];
      cell[d][0] = max(below, 0);
      cell[d][1] = min(half, nc[d] - 1);
      const double n0 = odd ? P.N[1][0] : P.N[2][0], n1 = odd ? P.N[1][1] : P.N[2][1];
      const double c0 = odd ? P.C[1][0] : P.C[2][0], c1 = odd ? P.C[1][1] : P.C[2][1];
      wN0[d][0] = v0 ? P.h[d] * n0 : 0.0;
      wN1[d][0] = v0 ? P.h[d] * n1 : 0.0;
      wC0[d][0] = v0 ? c0 : 0.0;
      wC1[d][0] = v0 ? c1 : 0.0;
      wN0[d][1] = v1 ? P.h[d] * P.N[0][0] : 0.0;
      wN1[d][1] = v1 ? P.h[d] * P.N[0][1] : 0.0;
      wC0[d][1] = v1 ? P.C[0][0] : 0.0;
      wC1[d][1] = v1 ? P.C[0][1] : 0.0;
    }
    _Pragma("unroll 1") for (int s = 0; s < P.nsrc; ++s) { // (a run-time loop: only the destination loops need compile-time bounds)
      double g[3] = {0, 0, 0};
#pragma unroll
      for (int ez = 0; ez < 2; ++ez)
#pragma unroll
        for (int ey = 0; ey < 2; ++ey)
#pragma unroll
          for (int ex = 0; ex < 2; ++ex) {
            const double *pc = P.p[s] + 4 * (cell[0][ex] + (long long)P.ncx * (cell[1][ey] + (long long)P.ncy * cell[2][ez]));
            const double q0 = pc[0], q1 = pc[1], q2 = pc[2], q3 = pc[3];
            const double Nx0 = wN0[0][ex], Nx1 = wN1[0][ex], Ny0 = wN0[1][ey], Ny1 = wN1[1][ey], Nz0 = wN0[2][ez], Nz1 = wN1[2][ez];
            const double Cx0 = wC0[0][ex], Cx1 = wC1[0][ex], Cy0 = wC0[1][ey], Cy1 = wC1[1][ey], Cz0 = wC0[2][ez], Cz1 = wC1[2][ez];
            // int (q0 + q1 l(xi) + q2 l(eta) + q3 l(zeta)) d phi / d x_c
            g[0] += (q0 * Cx0 + q1 * Cx1) * Ny0 * Nz0 + Cx0 * (q2 * Ny1 * Nz0 + q3 * Ny0 * Nz1);
            g[1] += (q0 * Cy0 + q2 * Cy1) * Nx0 * Nz0 + Cy0 * (q1 * Nx1 * Nz0 + q3 * Nx0 * Nz1);
            g[2] += (q0 * Cz0 + q3 * Cz1) * Nx0 * Ny0 + Cz0 * (q1 * Nx1 * Ny0 + q2 * Nx0 * Ny1);
          }
_Pragma("unroll") for (int o = 0; o < NO; ++o)
        if (o < P.nout)
          for (int c = 0; c < 3; ++c) acc[o][c] = fma(P.wKu[o][s], g[c], acc[o][c]);
    }
  }
_Pragma("unroll") for (int o = 0; o < NO; ++o)
    if (o < P.nout && P.out_u[o])
      for (int c = 0; c < 3; ++c) {
        if constexpr (NO <= 2) P.out_u[o][c * P.Nu + node] = old[o][c] - acc[o][c];
        else P.out_u[o][c * P.Nu + node] -= acc[o][c];
      }
}

// out_p[o] (=, +=) sum_s wKp[o][s] sum_c B_c u_s,c: one thread per pressure DoF (FE_Q(1) node / FE_DGP(1) cell function)
// (Five threads per FE_Q(1) node, one per z-plane of its 5 x 5 x 5 neighbourhood, with the partial sums added in LDS, measured
// slower: 79 against 62 us on 64^3 cells - the per-thread weight set-up is what one thread per node amortises.)
template <int NS, int NO, bool PDG>
__global__ __launch_bounds__(256, NO <= 2 ? 4 : 2) void stokes_div_kernel(const CouplingParams P, long long Np)
{
  const long long dof = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (dof >= Np) return;
  const int nc[3] = {P.ncx, P.ncy, P.ncz};
  double acc[NO];
#pragma unroll
  for (int o = 0; o < NO; ++o) acc[o] = 0.0;
  auto con = [&](int ix, int iy, int iz) {
    return ((P.dmask & 1) && ix == 0) || ((P.dmask & 2) && ix == P.ndu[0] - 1) || ((P.dmask & 4) && iy == 0) ||
           ((P.dmask & 8) && iy == P.ndu[1] - 1) || ((P.dmask & 16) && iz == 0) || ((P.dmask & 32) && iz == P.ndu[2] - 1);
  };
  if constexpr (!PDG) {
    const int j[3] = {int(dof % P.ndp[0]), int((dof / P.ndp[0]) % P.ndp[1]), int(dof / ((long long)P.ndp[0] * P.ndp[1]))};
    // velocity line nodes 2 j - 2 .. 2 j + 2: (cell j - 1: nodes 0, 1, 2 against psi_1), (cell j: nodes 0, 1, 2 against psi_0)
    double wn[3][5], wc[3][5];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const bool lo = j[d] > 0, hi = j[d] < nc[d];
      wn[d][0] = lo ? P.N[0][1] : 0.0; wn[d][1] = lo ? P.N[1][1] : 0.0; wn[d][2] = (lo ? P.N[2][1] : 0.0) + (hi ? P.N[0][0] : 0.0);
      wn[d][3] = hi ? P.N[1][0] : 0.0; wn[d][4] = hi ? P.N[2][0] : 0.0;
      wc[d][0] = lo ? P.C[0][1] : 0.0; wc[d][1] = lo ? P.C[1][1] : 0.0; wc[d][2] = (lo ? P.C[2][1] : 0.0) + (hi ? P.C[0][0] : 0.0);
      wc[d][3] = hi ? P.C[1][0] : 0.0; wc[d][4] = hi ? P.C[2][0] : 0.0;
#pragma unroll
      for (int k = 0; k < 5; ++k) wn[d][k] *= P.h[d];
    }
    // nodes beyond the lattice and constrained nodes (they read as 0) carry weight 0: the loops are free of branches
    const int lim[3] = {P.ndu[0] - 1, P.ndu[1] - 1, P.ndu[2] - 1};
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const int i = 2 * j[d] - 2 + k;
        const bool off = i < 0 || i > lim[d] || ((P.dmask >> (2 * d) & 1) && i == 0) || ((P.dmask >> (2 * d + 1) & 1) && i == lim[d]);
        if (off) wn[d][k] = wc[d][k] = 0.0;
      }
    // (the y / z loops stay rolled: their weights are picked with selects on the wave-uniform loop counters, not indexed)
    auto pick = [](const double (&w)[5], int k) { return k == 0 ? w[0] : (k == 1 ? w[1] : (k == 2 ? w[2] : (k == 3 ? w[3] : w[4]))); };
    int ixs[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) ixs[k] = min(max(2 * j[0] - 2 + k, 0), lim[0]);
    _Pragma("unroll 1") for (int s = 0; s < P.nsrc; ++s) {
      double dv = 0.0;
      const double *us = P.u[s];
      _Pragma("unroll 1") for (int kz = 0; kz < 5; ++kz) {
        const int iz = min(max(2 * j[2] - 2 + kz, 0), lim[2]);
        const double nz = pick(wn[2], kz), cz = pick(wc[2], kz);
        _Pragma("unroll 1") for (int ky = 0; ky < 5; ++ky) {
          const int iy = min(max(2 * j[1] - 2 + ky, 0), lim[1]);
          const double ny = pick(wn[1], ky), cy = pick(wc[1], ky);
          const double *row = us + (long long)P.ndu[0] * (iy + (long long)P.ndu[1] * iz);
          double sx = 0.0, sy = 0.0, sz = 0.0;
#pragma unroll
          for (int kx = 0; kx < 5; ++kx) {
            const double *uu = row + ixs[kx];
            sx = fma(wc[0][kx], uu[0], sx);
            sy = fma(wn[0][kx], uu[P.Nu], sy);
            sz = fma(wn[0][kx], uu[2 * P.Nu], sz);
          }
          dv = fma(ny * nz, sx, fma(cy * nz, sy, fma(ny * cz, sz, dv)));
        }
      }
      _Pragma("unroll") for (int o = 0; o < NO; ++o)
        if (o < P.nout) acc[o] = fma(P.wKp[o][s], dv, acc[o]);
    }
  } else {
    const long long cell = dof >> 2;
    const int fn = int(dof & 3);
    const int cx = int(cell % P.ncx), cy = int((cell / P.ncx) % P.ncy), cz = int(cell / ((long long)P.ncx * P.ncy));
    _Pragma("unroll 1") for (int s = 0; s < P.nsrc; ++s) { // (a run-time loop: only the destination loops need compile-time bounds)
      double dv = 0.0;
      for (int az = 0; az < 3; ++az)
        for (int ay = 0; ay < 3; ++ay)
          for (int ax = 0; ax < 3; ++ax) {
            const int ix = 2 * cx + ax, iy = 2 * cy + ay, iz = 2 * cz + az;
            if (con(ix, iy, iz)) continue;
            const double *uu = P.u[s] + (ix + (long long)P.ndu[0] * (iy + (long long)P.ndu[1] * iz));
            // test function fn: 1 / l(xi) / l(eta) / l(zeta): index 1 of N / C in that direction
            const int fx = fn == 1, fy = fn == 2, fz = fn == 3;
            const double Nx = P.h[0] * P.N[ax][fx], Ny = P.h[1] * P.N[ay][fy], Nz = P.h[2] * P.N[az][fz];
            dv = fma(P.C[ax][fx] * Ny * Nz, uu[0], dv);
            dv = fma(Nx * P.C[ay][fy] * Nz, uu[P.Nu], dv);
            dv = fma(Nx * Ny * P.C[az][fz], uu[2 * P.Nu], dv);
          }
      _Pragma("unroll") for (int o = 0; o < NO; ++o)
        if (o < P.nout) acc[o] = fma(P.wKp[o][s], dv, acc[o]);
    }
  }
  _Pragma("unroll") for (int o = 0; o < NO; ++o)
    if (o < P.nout && P.out_p[o]) {
      if (P.store_p[o]) P.out_p[o][dof] = acc[o];
      else P.out_p[o][dof] += acc[o];
    }
}

// FE_DGP(1): one thread per CELL computes the cell's four pressure rows from its 27 x 3 velocity values (the one-thread-per-DoF form
// above reads them four times: 157 us beside the sweep on 64^3 cells)
template <int NS, int NO>
__global__ __launch_bounds__(256, 2) void stokes_div_dgp_cell_kernel(const CouplingParams P, long long ncells)
{
  const long long cell = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= ncells) return;
  const int cx = int(cell % P.ncx), cy = int((cell / P.ncx) % P.ncy), cz = int(cell / ((long long)P.ncx * P.ncy));
  double acc[NO][4];
#pragma unroll
  for (int o = 0; o < NO; ++o)
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[o][f] = 0.0;
  _Pragma("unroll 1") for (int s = 0; s < P.nsrc; ++s) {
    double dv[4] = {0.0, 0.0, 0.0, 0.0};
    const double *us = P.u[s];
    _Pragma("unroll 1") for (int az = 0; az < 3; ++az) { // (the z and y loops stay rolled: nine loads in flight, a few dozen registers)
      const int iz = 2 * cz + az;
      const bool conz = ((P.dmask & 16) && iz == 0) || ((P.dmask & 32) && iz == P.ndu[2] - 1);
      const double Nz0 = P.h[2] * P.N[az][0], Nz1 = P.h[2] * P.N[az][1], Cz0 = P.C[az][0], Cz1 = P.C[az][1];
      _Pragma("unroll 1") for (int ay = 0; ay < 3; ++ay) {
        const int iy = 2 * cy + ay;
        const bool cony = conz || ((P.dmask & 4) && iy == 0) || ((P.dmask & 8) && iy == P.ndu[1] - 1);
        const double Ny0 = P.h[1] * P.N[ay][0], Ny1 = P.h[1] * P.N[ay][1], Cy0 = P.C[ay][0], Cy1 = P.C[ay][1];
        const double *row = us + (long long)P.ndu[0] * (iy + (long long)P.ndu[1] * iz) + 2 * cx;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
          const int ix = 2 * cx + ax;
          const bool con = cony || ((P.dmask & 1) && ix == 0) || ((P.dmask & 2) && ix == P.ndu[0] - 1);
          const double ux = con ? 0.0 : row[ax], uy = con ? 0.0 : row[P.Nu + ax], uz = con ? 0.0 : row[2 * P.Nu + ax];
          const double Nx0 = P.h[0] * P.N[ax][0], Nx1 = P.h[0] * P.N[ax][1], Cx0 = P.C[ax][0], Cx1 = P.C[ax][1];
          // test functions 1, l(xi), l(eta), l(zeta): index 1 of N / C in that direction
          dv[0] += Cx0 * Ny0 * Nz0 * ux + Nx0 * Cy0 * Nz0 * uy + Nx0 * Ny0 * Cz0 * uz;
          dv[1] += Cx1 * Ny0 * Nz0 * ux + Nx1 * Cy0 * Nz0 * uy + Nx1 * Ny0 * Cz0 * uz;
          dv[2] += Cx0 * Ny1 * Nz0 * ux + Nx0 * Cy1 * Nz0 * uy + Nx0 * Ny1 * Cz0 * uz;
          dv[3] += Cx0 * Ny0 * Nz1 * ux + Nx0 * Cy0 * Nz1 * uy + Nx0 * Ny0 * Cz1 * uz;
        }
      }
    }
#pragma unroll
    for (int o = 0; o < NO; ++o)
      if (o < P.nout)
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[o][f] = fma(P.wKp[o][s], dv[f], acc[o][f]);
  }
#pragma unroll
  for (int o = 0; o < NO; ++o)
    if (o < P.nout && P.out_p[o]) {
      double *q = P.out_p[o] + 4 * cell;
#pragma unroll
      for (int f = 0; f < 4; ++f) q[f] = P.store_p[o] ? acc[o][f] : q[f] + acc[o][f];
    }
}

// The same for FE_Q(1) as a MARCH along z: a thread takes DIV_SEG consecutive pressure nodes of a z-line and keeps, per velocity
// z-plane of its 5 x 5 (x, y) neighbourhood, the two partial sums the nodes above and below share (s1 = sum of the in-plane terms of
// the x and y components, s2 = of the z component): two new planes per node instead of five, 2.5 x fewer loads, and a twentieth of
// the threads - the kernel runs beside the velocity sweep, where every instruction it issues competes with the sweep's.
constexpr int DIV_SEG = 4;
template <int NS, int NO>
__global__ __launch_bounds__(256, 2) void stokes_div_march_kernel(const CouplingParams P, int nseg)
{
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long ncol = (long long)P.ndp[0] * P.ndp[1];
  if (t >= ncol * nseg) return;
  const int jx = int(t % P.ndp[0]), jy = int((t / P.ndp[0]) % P.ndp[1]), seg = int(t / ncol);
  const int nc[3] = {P.ncx, P.ncy, P.ncz};
  const int lim[3] = {P.ndu[0] - 1, P.ndu[1] - 1, P.ndu[2] - 1};
  const int jxy[2] = {jx, jy};
  // in-plane weights of this line: velocity line nodes 2 j - 2 .. 2 j + 2 (cell j - 1 against psi_1, cell j against psi_0); nodes beyond
  // the lattice and constrained nodes carry weight 0
  double wn[2][5], wc[2][5];
  int idx[2][5];
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const bool lo = jxy[d] > 0, hi = jxy[d] < nc[d];
    wn[d][0] = lo ? P.N[0][1] : 0.0; wn[d][1] = lo ? P.N[1][1] : 0.0; wn[d][2] = (lo ? P.N[2][1] : 0.0) + (hi ? P.N[0][0] : 0.0);
    wn[d][3] = hi ? P.N[1][0] : 0.0; wn[d][4] = hi ? P.N[2][0] : 0.0;
    wc[d][0] = lo ? P.C[0][1] : 0.0; wc[d][1] = lo ? P.C[1][1] : 0.0; wc[d][2] = (lo ? P.C[2][1] : 0.0) + (hi ? P.C[0][0] : 0.0);
    wc[d][3] = hi ? P.C[1][0] : 0.0; wc[d][4] = hi ? P.C[2][0] : 0.0;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int i = 2 * jxy[d] - 2 + k;
      const bool off = i < 0 || i > lim[d] || ((P.dmask >> (2 * d) & 1) && i == 0) || ((P.dmask >> (2 * d + 1) & 1) && i == lim[d]);
      wn[d][k] = off ? 0.0 : wn[d][k] * P.h[d];
      wc[d][k] = off ? 0.0 : wc[d][k];
      idx[d][k] = min(max(i, 0), lim[d]);
    }
  }
  // the two partial sums of velocity plane iz (0 beyond the lattice and on constrained planes)
  auto plane = [&](int s, int iz, double &s1, double &s2) {
    s1 = s2 = 0.0;
    const bool off = iz < 0 || iz > lim[2] || ((P.dmask & 16) && iz == 0) || ((P.dmask & 32) && iz == lim[2]);
    if (off) return; // (wave-uniform for a launch whose threads of a wave share the segment)
    const double *us = P.u[s] + (long long)P.ndu[0] * P.ndu[1] * iz;
    _Pragma("unroll 1") for (int ky = 0; ky < 5; ++ky) {
      const double *row = us + (long long)P.ndu[0] * idx[1][ky];
      const double ny = ky == 0 ? wn[1][0] : (ky == 1 ? wn[1][1] : (ky == 2 ? wn[1][2] : (ky == 3 ? wn[1][3] : wn[1][4])));
      const double cy = ky == 0 ? wc[1][0] : (ky == 1 ? wc[1][1] : (ky == 2 ? wc[1][2] : (ky == 3 ? wc[1][3] : wc[1][4])));
      double sx = 0.0, sy = 0.0, sz = 0.0;
#pragma unroll
      for (int kx = 0; kx < 5; ++kx) {
        const double *uu = row + idx[0][kx];
        sx = fma(wc[0][kx], uu[0], sx);
        sy = fma(wn[0][kx], uu[P.Nu], sy);
        sz = fma(wn[0][kx], uu[2 * P.Nu], sz);
      }
      s1 = fma(ny, sx, fma(cy, sy, s1));
      s2 = fma(ny, sz, s2);
    }
  };
  const int j0 = int((long long)P.ndp[2] * seg / nseg), j1 = int((long long)P.ndp[2] * (seg + 1) / nseg);
  double s1[NS][5], s2[NS][5]; // planes 2 j - 2 .. 2 j + 2 of the current node
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int k = 0; k < 5; ++k) s1[s][k] = s2[s][k] = 0.0;
  for (int j = j0; j < j1; ++j) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
      if (s < P.nsrc) {
        if (j == j0) {
#pragma unroll
          for (int k = 0; k < 3; ++k) plane(s, 2 * j - 2 + k, s1[s][k], s2[s][k]);
        }
        plane(s, 2 * j + 1, s1[s][3], s2[s][3]);
        plane(s, 2 * j + 2, s1[s][4], s2[s][4]);
      }
    const bool lo = j > 0, hi = j < nc[2];
    const double wnz[5] = {lo ? P.N[0][1] : 0.0, lo ? P.N[1][1] : 0.0, (lo ? P.N[2][1] : 0.0) + (hi ? P.N[0][0] : 0.0), hi ? P.N[1][0] : 0.0, hi ? P.N[2][0] : 0.0};
    const double wcz[5] = {lo ? P.C[0][1] : 0.0, lo ? P.C[1][1] : 0.0, (lo ? P.C[2][1] : 0.0) + (hi ? P.C[0][0] : 0.0), hi ? P.C[1][0] : 0.0, hi ? P.C[2][0] : 0.0};
    double acc[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) acc[o] = 0.0;
#pragma unroll
    for (int s = 0; s < NS; ++s)
      if (s < P.nsrc) {
        double dv = 0.0;
#pragma unroll
        for (int k = 0; k < 5; ++k) dv = fma(P.h[2] * wnz[k], s1[s][k], fma(wcz[k], s2[s][k], dv));
#pragma unroll
        for (int o = 0; o < NO; ++o)
          if (o < P.nout) acc[o] = fma(P.wKp[o][s], dv, acc[o]);
      }
    const long long dof = jx + (long long)P.ndp[0] * (jy + (long long)P.ndp[1] * j);
#pragma unroll
    for (int o = 0; o < NO; ++o)
      if (o < P.nout && P.out_p[o]) {
        if (P.store_p[o]) P.out_p[o][dof] = acc[o];
        else P.out_p[o][dof] += acc[o];
      }
    // the next node shares planes 2 j .. 2 j + 2
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int k = 0; k < 3; ++k) { s1[s][k] = s1[s][k + 2]; s2[s][k] = s2[s][k + 2]; }
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
  // axis-aligned uniform meshes: the scalar FE_Q(2) context whose pencil sweep applies nu K + wM M to the velocity components,
  // and the 1D tables of the coupling kernels
  stfem_ctx *scalar = nullptr;
  CouplingParams coupling;
  // the divergence kernel reads the sources and writes the pressure destinations only: it runs beside the velocity sweep on a
  // stream of the context's own, forked from and joined to the caller's stream with two events
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  stfem_ctx *pressure_space = nullptr; // scalar context behind the pressure vectors (stfem_stokes_pressure_ctx), made on demand
  double *d_pq = nullptr;              // exact values at the pressure quadrature points (stfem_stokes_pressure_difference)
  size_t pq_points = 0;
  double *d_pred = nullptr;            // its reduction results
  int weak_mask = 0, outflow_mask = 0;
  double penalty1 = 20.0, penalty2 = 10.0;
  BoundaryParams bnd;
  double *d_g = nullptr; // Dirichlet data at the face quadrature points (stfem_stokes_nitsche_rhs)
  size_t g_points = 0;
  std::vector<double> h_vertices;
};

static thread_local char g_stokes_err[256] = "";
static int stokes_lowest_priority()
{
  int lo = 0, hi = 0; // (numerically highest value = lowest priority)
  if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return lo;
}
// One side stream per device for all Stokes contexts (a multigrid has one context per level: a stream each would outnumber the
// hardware queues, and dependencies between streams that share a queue are resolved on the host).  Never destroyed.
static hipStream_t stokes_side_stream(int device)
{
  static std::mutex mu;
  static hipStream_t streams[64] = {};
  if (device < 0 || device >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (!streams[device] && hipStreamCreateWithPriority(&streams[device], hipStreamNonBlocking, stokes_lowest_priority()) != hipSuccess) {
    (void)hipGetLastError();
    streams[device] = nullptr;
  }
  return streams[device];
}
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
  std::memset(&c->coupling, 0, sizeof(c->coupling));
  if (b.cart) {
    const char *e = getenv("STFEM_STOKES_CELL"); // 1: keep the cell kernel on boxes too (cross-checks, measurements)
    if (!(e && atoi(e) != 0)) {
      stfem_mesh_desc md = *mesh;
      md.vertices = nullptr;
      stfem_space_desc sd{2, 3, 1, 0};
      const int rc = stfem_ctx_create(&md, &sd, &c->scalar);
      if (rc != STFEM_OK) c->scalar = nullptr; // (the cell kernel serves then)
      else if (!(c->side = stokes_side_stream(c->device)) || hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
               hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        c->side = nullptr; // (the divergence kernel then follows the sweep on the caller's stream)
      }
    }
    CouplingParams &k = c->coupling;
    k.ncx = c->nc[0]; k.ncy = c->nc[1]; k.ncz = c->nc[2];
    for (int d = 0; d < 3; ++d) {
      k.ndu[d] = c->ndu[d]; k.ndp[d] = c->ndp[d];
      k.h[d] = (mesh->upper[d] - mesh->lower[d]) / c->nc[d];
    }
    k.Nu = c->Nu; k.dmask = c->dmask; k.pdg = c->pspace;
    // 1D integrals on the reference cell with the operator's Gauss rule (exact for these polynomials)
    for (int a = 0; a < 3; ++a)
      for (int j = 0; j < 2; ++j) {
        double n = 0.0, cc = 0.0;
        for (int q = 0; q < 3; ++q) {
          const double psi = c->pspace ? (j == 0 ? 1.0 : std::sqrt(3.0) * (2.0 * xq[q] - 1.0)) : Sp[q * 2 + j];
          n += wq[q] * tu.S[q * 3 + a] * psi;
          cc += wq[q] * tu.D[q * 3 + a] * psi;
        }
        k.N[a][j] = n;
        k.C[a][j] = cc;
      }
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
  if (c->scalar) stfem_ctx_destroy(c->scalar);
  if (c->pressure_space) stfem_ctx_destroy(c->pressure_space);
  if (c->d_pq) (void)hipFree(c->d_pq);
  if (c->d_pred) (void)hipFree(c->d_pred);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  delete c;
}

} // extern "C"
int stfem_stokes_internal_desc(const stfem_stokes_ctx *c, stfem_stokes_desc *d)
{
  if (!c || !d) return STFEM_ERR_INVALID_ARGUMENT;
  d->device = c->device; d->cart = c->base.cart; d->pspace = c->pspace; d->dmask = c->dmask;
  d->weak_mask = c->weak_mask; d->outflow_mask = c->outflow_mask;
  for (int k = 0; k < 3; ++k) {
    d->nc[k] = c->nc[k]; d->ndu[k] = c->ndu[k]; d->ndp[k] = c->ndp[k];
    d->lower[k] = c->h_vertices[k];
    d->upper[k] = c->h_vertices[c->h_vertices.size() - 3 + k];
  }
  d->Nu = c->Nu; d->Np = c->Np; d->nu = c->nu; d->penalty1 = c->penalty1; d->penalty2 = c->penalty2;
  return STFEM_OK;
}
extern "C" {

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

// The Kronecker path of stokes_launch (axis-aligned uniform meshes): the same arguments, three steps.
static int stokes_cart_launch(stfem_stokes_ctx *c, const StokesParams &prm, hipStream_t st)
{
  const int nsrc = prm.nsrc > 1 ? prm.nsrc : 1, nout = prm.nout;
  if (nsrc > MAXSRC || nout > MAXOUT) return STFEM_ERR_UNSUPPORTED;
  const double *us[MAXSRC], *ps[MAXSRC];
  double wKu[MAXOUT][MAXSRC], wKp[MAXOUT][MAXSRC], wM[MAXOUT][MAXSRC];
  for (int q = 0; q < nsrc; ++q) {
    us[q] = prm.nsrc > 1 ? prm.us[q] : prm.u;
    ps[q] = prm.nsrc > 1 ? prm.ps[q] : prm.p;
    for (int o = 0; o < nout; ++o) {
      wKu[o][q] = prm.nsrc > 1 ? prm.fKu[o][q] : prm.wKu[o];
      wKp[o][q] = prm.nsrc > 1 ? prm.fKp[o][q] : prm.wKp[o];
      wM[o][q] = prm.nsrc > 1 ? prm.fM[o][q] : prm.wM[o];
    }
  }
  // ---- the coupling, in gather form: parameters
  bool k_u = false, k_p = false;
  for (int o = 0; o < nout; ++o)
    for (int q = 0; q < nsrc; ++q) {
      k_u = k_u || (prm.out_u[o] && wKu[o][q] != 0.0 && ps[q]);
      k_p = k_p || (prm.out_p[o] && wKp[o][q] != 0.0);
    }
  CouplingParams k = c->coupling;
  k.nsrc = nsrc; k.nout = nout;
  for (int q = 0; q < nsrc; ++q) { k.u[q] = us[q]; k.p[q] = ps[q]; }
  for (int o = 0; o < nout; ++o) {
    k.out_u[o] = prm.out_u[o]; k.out_p[o] = prm.out_p[o];
    k.store_p[o] = prm.store_p[o];
    for (int q = 0; q < nsrc; ++q) { k.wKu[o][q] = ps[q] ? wKu[o][q] : 0.0; k.wKp[o][q] = wKp[o][q]; }
  }
  (void)hipGetLastError();
#define STOKES_COUPLING_LAUNCH(KERN, NS_, NO_, GRID, ...)                                                          \
  do {                                                                                                             \
    if (k.pdg) hipLaunchKernelGGL((stokes_##KERN##_kernel<NS_, NO_, true>), dim3(GRID), dim3(256), 0, st, __VA_ARGS__);  \
    else hipLaunchKernelGGL((stokes_##KERN##_kernel<NS_, NO_, false>), dim3(GRID), dim3(GRID##_threads), 0, st, __VA_ARGS__); \
  } while (0)
  const unsigned gu = (unsigned)((c->Nu + 255) / 256), gp = (unsigned)((c->Np + 255) / 256);
  const unsigned gu_threads = 256, gp_threads = 256;
  const int shape = (nsrc == 1 && nout == 1) ? 0 : ((nsrc <= 2 && nout <= 2) ? 1 : ((nsrc <= MAXSRC && nout <= 4) ? 2 : 3));
  // ---- 2. out_p (=, +=) sum_q wKp B u_q, beside the velocity sweep (STFEM_STOKES_SERIAL=1: on the caller's stream, after it)
  static const bool serial = [] { const char *e = getenv("STFEM_STOKES_SERIAL"); return e && atoi(e) != 0; }();
  bool any_p = false;
  for (int o = 0; o < nout; ++o) any_p = any_p || prm.out_p[o];
  (void)k_p; // (a destination that is overwritten is written even with zero weights)
  // The fork onto the side stream - ONE per device, shared by every Stokes context: with a stream per context (a multigrid has one
  // context per level) the streams outnumbered the hardware queues, and the multigrid-preconditioned solve ran three times slower
  // than on one stream (profiles/r3/experiments.txt Z: 64^3 cells, 47 against 16 ms per FGMRES iteration).  No fork where the
  // gradient term rides in the sweep (FE_Q(1), one time dof): that sweep leaves no registers for a second kernel on the CU.
  // STFEM_STOKES_FORK_MIN_CELLS=<n>: fork on meshes of at least n cells only (measurements).
  static const long long fork_min_cells = [] {
    const char *e = getenv("STFEM_STOKES_FORK_MIN_CELLS");
    return e ? atoll(e) : 0ll;
  }();
  static const bool unfused_grad = [] { const char *e = getenv("STFEM_STOKES_GRAD_KERNEL"); return e && atoi(e) != 0; }();
  int n_store_u = 0, n_add_u = 0;
  for (int o = 0; o < nout; ++o)
    if (prm.out_u[o]) (prm.store_u[o] ? n_store_u : n_add_u)++;
  const bool gradient_in_sweep = nsrc == 1 && n_store_u == 1 && n_add_u == 0 && !c->pspace && ps[0] && k_u && !unfused_grad;
  const bool forked = any_p && !serial && c->side && !gradient_in_sweep && (long long)k.ncx * k.ncy * k.ncz >= fork_min_cells;
  static const bool div_gather = [] { const char *e = getenv("STFEM_STOKES_DIV_GATHER"); return e && atoi(e) != 0; }();
  auto launch_div = [&](hipStream_t st) {
    if (k.pdg && shape <= 1 && !div_gather) { // FE_DGP(1), up to two time dofs: one thread per cell
      const long long ncells = (long long)k.ncx * k.ncy * k.ncz;
      const unsigned g = (unsigned)((ncells + 255) / 256);
      if (shape == 0) hipLaunchKernelGGL((stokes_div_dgp_cell_kernel<1, 1>), dim3(g), dim3(256), 0, st, k, ncells);
      else hipLaunchKernelGGL((stokes_div_dgp_cell_kernel<2, 2>), dim3(g), dim3(256), 0, st, k, ncells);
      return;
    }
    if (!k.pdg && shape <= 1 && !div_gather) { // FE_Q(1), up to two time dofs: the march along z
      const int nseg = std::max(1, (k.ndp[2] + DIV_SEG - 1) / DIV_SEG);
      const long long nthreads = (long long)k.ndp[0] * k.ndp[1] * nseg;
      const unsigned g = (unsigned)((nthreads + 255) / 256);
      if (shape == 0) hipLaunchKernelGGL((stokes_div_march_kernel<1, 1>), dim3(g), dim3(256), 0, st, k, nseg);
      else hipLaunchKernelGGL((stokes_div_march_kernel<2, 2>), dim3(g), dim3(256), 0, st, k, nseg);
      return;
    }
    if (shape == 0) STOKES_COUPLING_LAUNCH(div, 1, 1, gp, k, c->Np);
    else if (shape == 1) STOKES_COUPLING_LAUNCH(div, 2, 2, gp, k, c->Np);
    else if (shape == 2) STOKES_COUPLING_LAUNCH(div, MAXSRC, 4, gp, k, c->Np);
    else STOKES_COUPLING_LAUNCH(div, MAXSRC, MAXOUT, gp, k, c->Np);
  };
  // (the fork point is here, before the sweep; the side stream's commands are enqueued after the sweep's so that the sweep's
  // persistent workgroups - exactly the resident number - are placed first and the divergence kernel fills what is left)
  if (forked) STOKES_TRY(hipEventRecord(c->ev_fork, st));
  bool grad_fused = false;
  // ---- 1. velocity blocks: out_u[o] (=, +=) sum_q (nu wKu K + wM M) u_q, component by component as scalar FE_Q(2) systems
  // (one launch with the three components as blocks when there is a single source and destination)
  for (int pass = 0; pass < 2; ++pass) { // destinations that are overwritten, then those that are accumulated into
    int rows[MAXOUT], nr = 0;
    for (int o = 0; o < nout; ++o)
      if (prm.out_u[o] && (prm.store_u[o] != 0) == (pass == 0)) rows[nr++] = o;
    if (nr == 0) continue;
    const int ncomp_blocks = (nr == 1 && nsrc == 1) ? 3 : 1; // components as blocks of one launch, or one launch per component
    for (int c0 = 0; c0 < 3; c0 += ncomp_blocks) {
      const int nbo = nr * ncomp_blocks, nbi = nsrc * ncomp_blocks;
      std::vector<void *> dptr(nbo), sptr(nbi);
      std::vector<double> a(size_t(nbo) * nbi, 0.0), b(size_t(nbo) * nbi, 0.0);
      for (int cc = 0; cc < ncomp_blocks; ++cc) {
        for (int r = 0; r < nr; ++r) dptr[cc * nr + r] = prm.out_u[rows[r]] + (c0 + cc) * c->Nu;
        for (int q = 0; q < nsrc; ++q) sptr[cc * nsrc + q] = const_cast<double *>(us[q]) + (c0 + cc) * c->Nu;
        for (int r = 0; r < nr; ++r)
          for (int q = 0; q < nsrc; ++q) {
            a[size_t(cc * nr + r) * nbi + cc * nsrc + q] = c->nu * wKu[rows[r]][q];
            b[size_t(cc * nr + r) * nbi + cc * nsrc + q] = wM[rows[r]][q];
          }
      }
      stfem_vec *vd = nullptr, *vs = nullptr;
      int rc = stfem_vector_wrap(c->scalar, nbo, dptr.data(), &vd);
      if (rc == STFEM_OK) rc = stfem_vector_wrap(c->scalar, nbi, sptr.data(), &vs);
      // one time dof, FE_Q(1) pressure, destination overwritten: the sweep adds - wKu B^T p to what it stores (SweepParams::gp) and
      // the gradient kernel below is not needed (it read and wrote the whole velocity destination again)
      static const bool unfused = [] { const char *e = getenv("STFEM_STOKES_GRAD_KERNEL"); return e && atoi(e) != 0; }();
      const bool fuse = ncomp_blocks == 3 && pass == 0 && !c->pspace && ps[0] && wKu[rows[0]][0] != 0.0 && !unfused;
      if (fuse && rc == STFEM_OK) {
        double w[3][2][3][2];
        for (int d = 0; d < 3; ++d)
          for (int a3 = 0; a3 < 3; ++a3)
            for (int j = 0; j < 2; ++j) {
              w[d][0][a3][j] = c->coupling.C[a3][j];
              w[d][1][a3][j] = c->coupling.h[d] * c->coupling.N[a3][j];
            }
        rc = stfem_internal_set_gradient(c->scalar, ps[0], w, -wKu[rows[0]][0]);
      }
      if (rc == STFEM_OK) rc = stfem_st_vmult(c->scalar, nbo, nbi, a.data(), b.data(), 0, pass, vd, vs, st);
      if (fuse) {
        if (rc == STFEM_OK && c->scalar->grad_applied) grad_fused = true;
        (void)stfem_internal_set_gradient(c->scalar, nullptr, nullptr, 0.0);
      }
      if (vd) stfem_vector_destroy(vd);
      if (vs) stfem_vector_destroy(vs);
      if (rc != STFEM_OK) {
        snprintf(g_stokes_err, sizeof(g_stokes_err), "velocity sweep: status %d (%s)", rc, stfem_last_hip_error());
        return rc;
      }
    }
  }
  if (forked) {
    STOKES_TRY(hipStreamWaitEvent(c->side, c->ev_fork, 0));
    launch_div(c->side);
    STOKES_TRY(hipEventRecord(c->ev_join, c->side));
  }
  // ---- 3. out_u -= sum_q wKu B^T p_q
  if (k_u && !grad_fused) {
    if (shape == 0) STOKES_COUPLING_LAUNCH(grad, 1, 1, gu, k);
    else if (shape == 1) STOKES_COUPLING_LAUNCH(grad, 2, 2, gu, k);
    else if (shape == 2) STOKES_COUPLING_LAUNCH(grad, MAXSRC, 4, gu, k);
    else STOKES_COUPLING_LAUNCH(grad, MAXSRC, MAXOUT, gu, k);
  }
  if (forked) STOKES_TRY(hipStreamWaitEvent(st, c->ev_join, 0));
  else if (any_p) launch_div(st);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_stokes_err, sizeof(g_stokes_err), "stokes coupling kernels: %s", hipGetErrorString(e));
    return STFEM_ERR_HIP;
  }
  return STFEM_OK;
}

static int stokes_launch(stfem_stokes_ctx *c, StokesParams &prm, hipStream_t st)
{
  (void)hipGetLastError();
  if (c->scalar && prm.nout <= MAXOUT && prm.nsrc <= MAXSRC) { // axis-aligned uniform mesh: Kronecker path
    const int rc = stokes_cart_launch(c, prm, st);
    if (rc != STFEM_OK) return rc;
  } else
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
    const void *kerns[8] = {(const void *)stokes_cell_kernel<false, false, false>, (const void *)stokes_cell_kernel<true, false, false>,
                            (const void *)stokes_cell_kernel<false, true, false>,  (const void *)stokes_cell_kernel<true, true, false>,
                            (const void *)stokes_cell_kernel<false, false, true>,  (const void *)stokes_cell_kernel<true, false, true>,
                            (const void *)stokes_cell_kernel<false, true, true>,   (const void *)stokes_cell_kernel<true, true, true>};
    const int which = (prm.pdg ? 4 : 0) + (prm.nsrc > 1 ? 2 : 0) + (prm.cart ? 1 : 0);
    const void *kern = kerns[which];
    // resident workgroups per CU of the instantiations, asked once (not on the launch path)
    static int resident_of[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int &resident = resident_of[which];
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

// ---- the pressure space by itself: what the solver around the operator needs of it (tests/tp_03stokes.cc:404-425, 1047-1062,
// include/exact_solution.h:503-649, the pressure transfer of the Stokes multigrid levels) ----
extern "C++" {
namespace {
// sum JxW (p_h - p)^2 and max |p_h - p| over QGauss(nq)^3 of the cells; blockIdx.x = cell (axis-aligned uniform cells)
template <bool PDG>
__global__ __launch_bounds__(64) void pressure_difference_kernel(int ncx, int ncy, int ncz, int nq, double vol, const double *__restrict__ xq,
                                                                 const double *__restrict__ wq, const double *__restrict__ p,
                                                                 const double *__restrict__ exact, double *__restrict__ out)
{
  const long long cell = blockIdx.x;
  const int cx = int(cell % ncx), cy = int((cell / ncx) % ncy), cz = int(cell / ((long long)ncx * ncy));
  const int nq3 = nq * nq * nq;
  double l2 = 0.0, l8 = 0.0;
  for (int q = threadIdx.x; q < nq3; q += 64) {
    const int qx = q % nq, qy = (q / nq) % nq, qz = q / (nq * nq);
    const double x = xq[qx], y = xq[qy], z = xq[qz];
    double ph;
    if (PDG) {
      const double *c = p + 4 * cell;
      const double s3 = 1.7320508075688772;
      ph = c[0] + s3 * (c[1] * (2 * x - 1) + c[2] * (2 * y - 1) + c[3] * (2 * z - 1));
    } else {
      const int npx = ncx + 1, npy = ncy + 1;
      ph = 0.0;
      for (int k = 0; k < 2; ++k)
        for (int j = 0; j < 2; ++j)
          for (int i = 0; i < 2; ++i)
            ph += (i ? x : 1 - x) * (j ? y : 1 - y) * (k ? z : 1 - z) * p[(cx + i) + (long long)npx * ((cy + j) + (long long)npy * (cz + k))];
    }
    const double e = ph - exact[cell * nq3 + q];
    l2 += vol * wq[qx] * wq[qy] * wq[qz] * e * e;
    l8 = fmax(l8, fabs(e));
  }
  __shared__ double s2[64], s8[64];
  s2[threadIdx.x] = l2; s8[threadIdx.x] = l8;
  __syncthreads();
  if (threadIdx.x == 0) { // fixed order: reproducible
    double a = 0.0, b = 0.0;
    for (int t = 0; t < 64; ++t) { a += s2[t]; b = fmax(b, s8[t]); }
    out[2 * cell] = a;
    out[2 * cell + 1] = b;
  }
}
__global__ __launch_bounds__(256) void pressure_difference_finish(long long ncells, const double *__restrict__ part, double *__restrict__ out)
{
  __shared__ double s2[256], s8[256];
  double a = 0.0, b = 0.0;
  for (long long c = threadIdx.x; c < ncells; c += 256) { a += part[2 * c]; b = fmax(b, part[2 * c + 1]); }
  s2[threadIdx.x] = a; s8[threadIdx.x] = b;
  __syncthreads();
  if (threadIdx.x == 0) {
    double x = 0.0, y = 0.0;
    for (int t = 0; t < 256; ++t) { x += s2[t]; y = fmax(y, s8[t]); }
    out[0] = x; out[1] = y;
  }
}
// FE_DGP(1) between a mesh and the mesh of its 2 x 2 x 2 children: the parent's function on child (sx, sy, sz) has the coefficients
// c0 + sqrt 3 ((sx - 1/2) c1 + (sy - 1/2) c2 + (sz - 1/2) c3), c1 / 2, c2 / 2, c3 / 2 (the embedding MGTwoLevelTransfer prolongates
// with; its restriction is the transpose).  One thread per coarse cell.
template <bool RESTRICT>
__global__ __launch_bounds__(256) void dgp_transfer_kernel(int ncx, int ncy, int ncz, double *__restrict__ dst, const double *__restrict__ src, int add)
{
  const long long cc = (long long)blockIdx.x * 256 + threadIdx.x; // coarse cell
  if (cc >= (long long)ncx * ncy * ncz) return;
  const int cx = int(cc % ncx), cy = int((cc / ncx) % ncy), cz = int(cc / ((long long)ncx * ncy));
  const int fx = 2 * ncx, fy = 2 * ncy;
  const double s3h = 0.8660254037844386; // sqrt 3 / 2
  double acc[4] = {0, 0, 0, 0};
  double pc[4] = {0, 0, 0, 0};
  if (!RESTRICT)
    for (int j = 0; j < 4; ++j) pc[j] = src[4 * cc + j];
  for (int sz = 0; sz < 2; ++sz)
    for (int sy = 0; sy < 2; ++sy)
      for (int sx = 0; sx < 2; ++sx) {
        const long long fc = (2 * cx + sx) + (long long)fx * ((2 * cy + sy) + (long long)fy * (2 * cz + sz));
        const double ox = sx ? s3h : -s3h, oy = sy ? s3h : -s3h, oz = sz ? s3h : -s3h;
        if (RESTRICT) {
          const double *f = src + 4 * fc;
          acc[0] += f[0];
          acc[1] += ox * f[0] + 0.5 * f[1];
          acc[2] += oy * f[0] + 0.5 * f[2];
          acc[3] += oz * f[0] + 0.5 * f[3];
        } else {
          double *f = dst + 4 * fc;
          const double v[4] = {pc[0] + ox * pc[1] + oy * pc[2] + oz * pc[3], 0.5 * pc[1], 0.5 * pc[2], 0.5 * pc[3]};
          for (int j = 0; j < 4; ++j) f[j] = add ? f[j] + v[j] : v[j];
        }
      }
  if (RESTRICT)
    for (int j = 0; j < 4; ++j) dst[4 * cc + j] = add ? dst[4 * cc + j] + acc[j] : acc[j];
}
} // namespace
} // extern "C++"

extern "C" {

// The scalar context behind the pressure vectors, for their vector arithmetic (stfem_vector_wrap + stfem_vector_axpby / stfem_dot /
// stfem_multi_dot ...) and, for FE_Q(1), for everything a FE_Q(1) function has in this library (load vectors, transfers, error norms).
// FE_Q(1): a degree-1 context on the mesh, no constraints.  FE_DGP(1): the arrays have 4 n_cells entries, which no mesh of this
// library's continuous elements has in general: the context is a CARRIER - a degree-1 context on 1 x 1 x (n_cells - 1) cells, i.e. with
// 2 x 2 x n_cells DoFs - good for the vector arithmetic only.  Owned by the Stokes context.
int stfem_stokes_pressure_ctx(stfem_stokes_ctx *c, stfem_ctx **out)
{
  if (!c || !out) return STFEM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (!c->pressure_space) {
    stfem_mesh_desc md;
    std::memset(&md, 0, sizeof(md));
    md.device = c->device;
    md.dirichlet_mask = 0;
    const long long ncells = (long long)c->nc[0] * c->nc[1] * c->nc[2];
    if (c->pspace) {
      if (ncells < 2 || ncells - 1 > 0x7fffffffll) return STFEM_ERR_UNSUPPORTED;
      md.ncell[0] = md.ncell[1] = 1;
      md.ncell[2] = int32_t(ncells - 1);
      for (int d = 0; d < 3; ++d) { md.lower[d] = 0.0; md.upper[d] = 1.0; }
    } else {
      for (int d = 0; d < 3; ++d) {
        md.ncell[d] = c->nc[d];
        md.lower[d] = c->h_vertices[d];
        md.upper[d] = c->h_vertices[c->h_vertices.size() - 3 + d];
      }
      if (!c->base.cart) md.vertices = c->h_vertices.data();
    }
    stfem_space_desc sd{1, 2, 1, 0};
    const int rc = stfem_ctx_create(&md, &sd, &c->pressure_space);
    if (rc != STFEM_OK) return rc;
  }
  *out = c->pressure_space;
  return STFEM_OK;
}

// The constant function and the mean-value functional of the pressure space (host arrays of n_pressure_dofs entries): ones = the
// coefficients of p = 1, weights = (1, psi_j) so that mean(p) = weights . p / volume (VectorTools::compute_mean_value /
// add_constant, tests/tp_03stokes.cc:1047-1062).  Axis-aligned uniform meshes.
int stfem_stokes_pressure_mean_vectors(stfem_stokes_ctx *c, double *ones, double *weights, double *volume)
{
  if (!c || !ones || !weights || !volume) return STFEM_ERR_INVALID_ARGUMENT;
  if (!c->base.cart) return STFEM_ERR_UNSUPPORTED;
  const double cell = c->base.detJ;
  const long long ncells = (long long)c->nc[0] * c->nc[1] * c->nc[2];
  *volume = cell * double(ncells);
  if (c->pspace) { // psi_0 = 1, the others have zero mean on a box
    for (long long i = 0; i < c->Np; ++i) { ones[i] = (i & 3) == 0 ? 1.0 : 0.0; weights[i] = (i & 3) == 0 ? cell : 0.0; }
  } else {
    for (int k = 0; k < c->ndp[2]; ++k)
      for (int j = 0; j < c->ndp[1]; ++j)
        for (int i = 0; i < c->ndp[0]; ++i) {
          const double wx = (i == 0 || i == c->ndp[0] - 1) ? 0.5 : 1.0, wy = (j == 0 || j == c->ndp[1] - 1) ? 0.5 : 1.0,
                       wz = (k == 0 || k == c->ndp[2] - 1) ? 0.5 : 1.0;
          const long long o = i + (long long)c->ndp[0] * (j + (long long)c->ndp[1] * k);
          ones[o] = 1.0;
          weights[o] = cell * wx * wy * wz;
        }
  }
  return STFEM_OK;
}

// quadrature points of QGauss(nq)^3 on the cells, out[cell][q][3], q = qx + nq (qy + nq qz) (axis-aligned uniform meshes)
int stfem_stokes_pressure_quadrature_points(const stfem_stokes_ctx *c, int nq, double *out)
{
  if (!c || !out || nq < 1 || nq > 8) return STFEM_ERR_INVALID_ARGUMENT;
  if (!c->base.cart) return STFEM_ERR_UNSUPPORTED;
  std::vector<double> xq, wq;
  stfem::gauss_rule(nq, xq, wq);
  double lo[3], h[3];
  for (int d = 0; d < 3; ++d) { lo[d] = c->h_vertices[d]; h[d] = 1.0 / c->base.hinv[d]; }
  size_t o = 0;
  for (int cz = 0; cz < c->nc[2]; ++cz)
    for (int cy = 0; cy < c->nc[1]; ++cy)
      for (int cx = 0; cx < c->nc[0]; ++cx)
        for (int qz = 0; qz < nq; ++qz)
          for (int qy = 0; qy < nq; ++qy)
            for (int qx = 0; qx < nq; ++qx, o += 3) {
              out[o] = lo[0] + h[0] * (cx + xq[qx]);
              out[o + 1] = lo[1] + h[1] * (cy + xq[qy]);
              out[o + 2] = lo[2] + h[2] * (cz + xq[qz]);
            }
  return STFEM_OK;
}

// out = { sum JxW (p_h - p)^2, max |p_h - p| } over those points (VectorTools::integrate_difference, L2_norm squared and Linfty_norm);
// p: device, exact_at_points: host [cell][q].  Synchronous.
int stfem_stokes_pressure_difference(stfem_stokes_ctx *c, int nq, const double *p, const double *exact_at_points, double out[2], void *stream)
{
  if (!c || !p || !exact_at_points || !out || nq < 1 || nq > 8) return STFEM_ERR_INVALID_ARGUMENT;
  if (!c->base.cart) return STFEM_ERR_UNSUPPORTED;
  STOKES_TRY(hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long ncells = (long long)c->nc[0] * c->nc[1] * c->nc[2];
  const size_t npts = size_t(ncells) * nq * nq * nq;
  if (c->pq_points < npts) {
    if (c->d_pq) STOKES_TRY(hipFree(c->d_pq));
    if (c->d_pred) STOKES_TRY(hipFree(c->d_pred));
    c->d_pq = c->d_pred = nullptr;
    c->pq_points = 0;
    if (hipMalloc(&c->d_pq, (npts + 16) * sizeof(double)) != hipSuccess || hipMalloc(&c->d_pred, (2 * size_t(ncells) + 2) * sizeof(double)) != hipSuccess)
      return STFEM_ERR_OUT_OF_MEMORY;
    c->pq_points = npts;
  }
  std::vector<double> xq, wq;
  stfem::gauss_rule(nq, xq, wq);
  std::vector<double> rule(xq);
  rule.insert(rule.end(), wq.begin(), wq.end());
  STOKES_TRY(hipMemcpyAsync(c->d_pq, exact_at_points, npts * sizeof(double), hipMemcpyHostToDevice, st));
  STOKES_TRY(hipMemcpyAsync(c->d_pq + npts, rule.data(), rule.size() * sizeof(double), hipMemcpyHostToDevice, st));
  (void)hipGetLastError();
  if (c->pspace)
    hipLaunchKernelGGL(pressure_difference_kernel<true>, dim3((unsigned)ncells), dim3(64), 0, st, c->nc[0], c->nc[1], c->nc[2], nq, c->base.detJ,
                       c->d_pq + npts, c->d_pq + npts + nq, p, c->d_pq, c->d_pred);
  else
    hipLaunchKernelGGL(pressure_difference_kernel<false>, dim3((unsigned)ncells), dim3(64), 0, st, c->nc[0], c->nc[1], c->nc[2], nq, c->base.detJ,
                       c->d_pq + npts, c->d_pq + npts + nq, p, c->d_pq, c->d_pred);
  hipLaunchKernelGGL(pressure_difference_finish, dim3(1), dim3(256), 0, st, ncells, c->d_pred, c->d_pred + 2 * ncells);
  if (hipGetLastError() != hipSuccess) return STFEM_ERR_HIP;
  STOKES_TRY(hipMemcpyAsync(out, c->d_pred + 2 * ncells, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
  STOKES_TRY(hipStreamSynchronize(st));
  return STFEM_OK;
}

// The FE_DGP(1) pressure between a mesh and the mesh with twice the cells per direction (the pressure variable's MGTwoLevelTransfer
// of the Stokes multigrid levels, include/stmg.h:557-600): prolongate: fine (=, +=) embedding of coarse; restrict: coarse (=, +=) its
// transpose applied to fine.  FE_Q(1) pressures use stfem_transfer_* on stfem_stokes_pressure_ctx.
int stfem_stokes_dgp_prolongate(stfem_stokes_ctx *fine, stfem_stokes_ctx *coarse, double *dst_fine, const double *src_coarse, int add, void *stream)
{
  if (!fine || !coarse || !dst_fine || !src_coarse) return STFEM_ERR_INVALID_ARGUMENT;
  if (!fine->pspace || !coarse->pspace) return STFEM_ERR_UNSUPPORTED;
  for (int d = 0; d < 3; ++d)
    if (fine->nc[d] != 2 * coarse->nc[d]) return STFEM_ERR_SHAPE_MISMATCH;
  STOKES_TRY(hipSetDevice(fine->device));
  const long long nc = (long long)coarse->nc[0] * coarse->nc[1] * coarse->nc[2];
  hipLaunchKernelGGL(dgp_transfer_kernel<false>, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), coarse->nc[0],
                     coarse->nc[1], coarse->nc[2], dst_fine, src_coarse, add);
  return hipGetLastError() == hipSuccess ? STFEM_OK : STFEM_ERR_HIP;
}
int stfem_stokes_dgp_restrict(stfem_stokes_ctx *fine, stfem_stokes_ctx *coarse, double *dst_coarse, const double *src_fine, int add, void *stream)
{
  if (!fine || !coarse || !dst_coarse || !src_fine) return STFEM_ERR_INVALID_ARGUMENT;
  if (!fine->pspace || !coarse->pspace) return STFEM_ERR_UNSUPPORTED;
  for (int d = 0; d < 3; ++d)
    if (fine->nc[d] != 2 * coarse->nc[d]) return STFEM_ERR_SHAPE_MISMATCH;
  STOKES_TRY(hipSetDevice(fine->device));
  const long long nc = (long long)coarse->nc[0] * coarse->nc[1] * coarse->nc[2];
  hipLaunchKernelGGL(dgp_transfer_kernel<true>, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), coarse->nc[0],
                     coarse->nc[1], coarse->nc[2], dst_coarse, src_fine, add);
  return hipGetLastError() == hipSuccess ? STFEM_OK : STFEM_ERR_HIP;
}

} // extern "C"
