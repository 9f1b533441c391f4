// Shared device code of the fused space-time cell sweep: wave geometry, the per-cell core
// (evaluate -> temporal combination -> quadrature-space operator -> integrate) and the
// Dirichlet masks.  Included by stfem_kernels.hip (atomic variant) and stfem_tile.hip.
//
// The core computes, for every cell and all temporal blocks at once,
//     out_j = sum_i alpha(j,i) K_cell u_i + beta(j,i) M_cell u_i
// i.e. the per-cell body of SystemMatrix::vmult (reference include/operators.h:536-559) around
// MatrixFreeOperator::do_cell_integral_local (operators.h:1135-1173), restructured: the temporal
// combination commutes with the spatial transforms, so the K and M parts of all blocks share one
// pipeline (Cartesian cells: 7 one-dimensional sweeps per output block, see cell_core; general
// cells: 13, see cell_core_general; the reference needs 2 x 12 per input block).
#pragma once
#include "stfem_device.h"
#include "stfem_kernels.h"

#include <hip/hip_runtime.h>

namespace stfem {
namespace STFEM_PREC {

template <int P, int NBM> struct Geometry {
  static constexpr int N = P + 1;
  static constexpr int CB_PER_WAVE = 64 / N;            // cell-blocks (cell x temporal block) per wave
  static constexpr int CELLS_PER_WAVE = CB_PER_WAVE / NBM;
  static constexpr int ACTIVE = CELLS_PER_WAVE * NBM * N; // active lanes
  static constexpr int CBS = N * N * N;                   // LDS doubles per cell-block
  static constexpr int WAVES = 4;
  static constexpr int LDS_PER_WAVE = CELLS_PER_WAVE * NBM * CBS;
};

// One pass of the fused operator over the cells owned by this wave (Cartesian cells, coefficient
// constant in the cell).
// PA: on entry the nodal src plane (layout A, [y][x]) of (cell, input block blk, z-plane k),
//     on exit the nodal result plane of (cell, output block blk, z-plane k).
//
// With exact Gauss quadrature the cell matrices are Kronecker products of the 1D nodal mass and
// stiffness matrices M1, K1, which are diagonalised simultaneously (K1 = W^T Lam W, M1 = W^T W):
//   sum_i (aK_i K_cell + aM_i M_cell) u_i
//     = (W^T x W^T x W^T) [ sum_i (aK_i (lx_a + ly_b + lz_c) + aM_i) (W x W x W) u_i ]
// i.e. three transforms to modal space, a diagonal scaling that also carries the temporal
// combination, three transforms back: 7 one-dimensional sweeps (8 for more than two blocks) and
// two wave-private transposes, against 10 sweeps and four transposes of the quadrature form
// of do_cell_integral_local (operators.h:1135-1173); identical up to rounding.
template <int P, int NBM>
__device__ __forceinline__ void
cell_core(const SweepParams &prm, real_t *__restrict__ lds, int cell_in_wave, int blk, int k,
          bool in_active, bool out_active, const real_t (&aK)[NBM], const real_t (&aM)[NBM],
          real_t (&PA)[(P + 1) * (P + 1)])
{
  using G = Geometry<P, NBM>;
  constexpr int N = G::N;
  constexpr int CBS = G::CBS;
  real_t *cb_lds = lds + (cell_in_wave * NBM + blk) * CBS;
#ifdef STFEM_ABLATION
  const bool no_lds = prm.experiment & 256;
#else
  constexpr bool no_lds = false;
#endif
  if (no_lds) { in_active = false; out_active = false; }

  // ---- phase A: to modal space in x, y (registers), hand over to layout B
  fd_plane<N, true, true>(prm.fd_W, PA);
  fd_plane<N, true, false>(prm.fd_W, PA);
  if (in_active) {
    STFEM_UNROLL
    for (int y = 0; y < N; ++y)
      STFEM_UNROLL
    for (int x = 0; x < N; ++x) cb_lds[k * N * N + y * N + x] = PA[y * N + x];
  }
  wave_lds_fence();

  // ---- phase B (this lane: x-mode k, registers [y-mode][z]): row by row, to modal space in z,
  // diagonal scaling + temporal combination, back to nodal values in z
  real_t lxk = prm.fd_lx[0];
  STFEM_UNROLL
  for (int m = 1; m < N; ++m) lxk = k == m ? prm.fd_lx[m] : lxk;
  // The diagonal factors aK (lx + ly + lz) + aM do not depend on the data; without these opaque
  // copies the compiler computes all NBM x 25 of them once, outside the caller's layer loop, and
  // keeps them in registers for the whole kernel.
  real_t wK[NBM], wM[NBM];
  STFEM_UNROLL
  for (int i = 0; i < NBM; ++i) {
    wK[i] = aK[i];
    wM[i] = aM[i];
    asm volatile("" : "+v"(wK[i]), "+v"(wM[i]));
  }
  asm volatile("" : "+v"(lxk));
  real_t R[N * N];
  // the rows of the input blocks are read one row ahead of their use; the laundered pointer and
  // the scheduling barrier keep the compiler from hoisting all rows' LDS reads to the top
  // (NBM x 25 extra registers)
  real_t vbuf[2][NBM][N];
  auto load_row = [&](int y, real_t (&v)[NBM][N]) {
    // (an opaque OFFSET: laundering the pointer itself would lose its LDS address space and turn
    // the reads into flat loads, which also wait for every global store in flight)
    int base = cell_in_wave * NBM * CBS + y * N + k;
    asm volatile("" : "+v"(base));
    STFEM_UNROLL
    for (int i = 0; i < NBM; ++i)
      if (i == 0 || i < prm.nbi) {
        STFEM_UNROLL
        for (int z = 0; z < N; ++z) v[i][z] = no_lds ? real_t(1) + z : lds[base + i * CBS + z * N * N];
      }
  };
  load_row(0, vbuf[0]);
  STFEM_UNROLL
  for (int y = 0; y < N; ++y) {
    if (y + 1 < N) load_row(y + 1, vbuf[(y + 1) & 1]);
    const real_t sy = lxk + prm.fd_ly[y];
    real_t acc[N];
    if (NBM <= 2) {
      // transform every input block, combine in modal space
      STFEM_UNROLL
      for (int i = 0; i < NBM; ++i) {
        if (i == 0 || i < prm.nbi) {
          real_t t[N];
          fd_forward<N>(prm.fd_W, vbuf[y & 1][i], t);
          STFEM_UNROLL
          for (int z = 0; z < N; ++z) {
            const real_t d = fma(wK[i], sy + prm.fd_lz[z], wM[i]);
            acc[z] = i == 0 ? d * t[z] : fma(d, t[z], acc[z]);
          }
        }
      }
    } else {
      // combine first (two combinations), transform both
      real_t ua[N], ub[N];
      STFEM_UNROLL
      for (int z = 0; z < N; ++z) ua[z] = ub[z] = real_t(0);
      STFEM_UNROLL
      for (int i = 0; i < NBM; ++i) {
        if (i < prm.nbi) {
          STFEM_UNROLL
          for (int z = 0; z < N; ++z) {
            ua[z] = fma(wK[i], vbuf[y & 1][i][z], ua[z]);
            ub[z] = fma(wM[i], vbuf[y & 1][i][z], ub[z]);
          }
        }
      }
      real_t ta[N], tb[N];
      fd_forward<N>(prm.fd_W, ua, ta);
      fd_forward<N>(prm.fd_W, ub, tb);
      STFEM_UNROLL
      for (int z = 0; z < N; ++z) acc[z] = fma(sy + prm.fd_lz[z], ta[z], tb[z]);
    }
    real_t r[N];
    fd_backward<N>(prm.fd_W, acc, r);
    pin(r);
    STFEM_UNROLL
    for (int z = 0; z < N; ++z) R[y * N + z] = r[z];
    __builtin_amdgcn_sched_barrier(0);
  }
  pin(R);
  wave_lds_fence();
  if (out_active) {
    STFEM_UNROLL
    for (int y = 0; y < N; ++y)
      STFEM_UNROLL
    for (int z = 0; z < N; ++z) cb_lds[z * N * N + y * N + k] = R[y * N + z];
  }
  wave_lds_fence();

  // ---- phase A2: back to nodal values in y, x
  STFEM_UNROLL
  for (int y = 0; y < N; ++y)
    STFEM_UNROLL
  for (int x = 0; x < N; ++x) PA[y * N + x] = no_lds ? R[y * N + x] : cb_lds[k * N * N + y * N + x];
  fd_plane<N, false, false>(prm.fd_W, PA);
  fd_plane<N, false, true>(prm.fd_W, PA);
  wave_lds_fence();
}

// General-geometry variant of cell_core (MappingQ1 cells of any shape, per-quadrature-point
// coefficients): the reference's evaluate -> quadrature loop -> integrate
// (include/operators.h:1135-1173) with unweighted S and collocation derivative D; the metric
// terms  G = c_L w detJ J^-1 J^-T (6 entries)  and  Mq = c_M w detJ  come precomputed per
// quadrature point from HBM (as deal.II's MatrixFree stores them), `met` = this cell's
// [qz][qy][qx][8] block: one 64-byte record (Gxx,Gxy,Gxz,Gyy,Gyz,Gzz,Mq,pad) per point, read
// with four 16-byte loads.  Six wave-private transposes instead of four: the x derivative lives in
// layout A, the flux contraction needs all three gradient components at one point (layout B).
template <int P, int NBM>
__device__ __forceinline__ void
cell_core_general(const SweepParams &prm, real_t *__restrict__ lds, int cell_in_wave, int blk, int k,
                  bool in_active, bool out_active, const real_t (&aK)[NBM], const real_t (&aM)[NBM],
                  const real_t *__restrict__ met, real_t (&PA)[(P + 1) * (P + 1)])
{
  using G = Geometry<P, NBM>;
  constexpr int N = G::N;
  constexpr int CBS = G::CBS;
  real_t *cb_lds = lds + (cell_in_wave * NBM + blk) * CBS;

  // ---- A: interpolate x, y
  plane_sweep<N, +1, true>(prm.eo_S, PA);
  plane_sweep<N, +1, false>(prm.eo_S, PA);
  if (in_active) {
    STFEM_UNROLL
    for (int y = 0; y < N; ++y)
      STFEM_UNROLL
    for (int x = 0; x < N; ++x) cb_lds[k * N * N + y * N + x] = PA[y * N + x];
  }
  wave_lds_fence();

  // ---- B1: temporal combination + interpolate z: Ua (Laplace part), R (mass part) at the
  // quadrature points of this lane's x-plane
  real_t Ua[N * N], R[N * N];
  STFEM_UNROLL
  for (int y = 0; y < N; ++y) {
    real_t ua[N], ub[N];
    STFEM_UNROLL
    for (int z = 0; z < N; ++z) ua[z] = ub[z] = real_t(0);
    STFEM_UNROLL
    for (int i = 0; i < NBM; ++i) {
      if (i < prm.nbi) {
        const real_t *in_lds = lds + (cell_in_wave * NBM + i) * CBS;
        STFEM_UNROLL
        for (int z = 0; z < N; ++z) {
          const real_t v = in_lds[z * N * N + y * N + k];
          ua[z] = fma(aK[i], v, ua[z]);
          ub[z] = fma(aM[i], v, ub[z]);
        }
      }
    }
    real_t ta[N], tb[N];
    eo_apply<N, +1>(prm.eo_S, ua, ta);
    eo_apply<N, +1>(prm.eo_S, ub, tb);
    STFEM_UNROLL
    for (int z = 0; z < N; ++z) {
      Ua[y * N + z] = ta[z];
      R[y * N + z] = tb[z];
    }
  }
  wave_lds_fence();
  if (out_active) {
    STFEM_UNROLL
    for (int y = 0; y < N; ++y)
      STFEM_UNROLL
    for (int z = 0; z < N; ++z) cb_lds[z * N * N + y * N + k] = Ua[y * N + z];
  }
  wave_lds_fence();

  // ---- A2: reference x derivative in layout A, back to layout B
  STFEM_UNROLL
  for (int y = 0; y < N; ++y)
    STFEM_UNROLL
  for (int x = 0; x < N; ++x) PA[y * N + x] = cb_lds[k * N * N + y * N + x];
  plane_sweep<N, -1, true>(prm.eo_Dq, PA);
  wave_lds_fence();
  if (out_active) {
    STFEM_UNROLL
    for (int y = 0; y < N; ++y)
      STFEM_UNROLL
    for (int x = 0; x < N; ++x) cb_lds[k * N * N + y * N + x] = PA[y * N + x];
  }
  wave_lds_fence();

  // ---- B3: y derivative (whole plane), then point by point: z derivative, metric, fluxes
  real_t Gy[N * N];
  STFEM_UNROLL
  for (int z = 0; z < N; ++z) {
    real_t x[N], y[N];
    STFEM_UNROLL
    for (int i = 0; i < N; ++i) x[i] = Ua[i * N + z];
    eo_apply<N, -1>(prm.eo_Dq, x, y);
    STFEM_UNROLL
    for (int i = 0; i < N; ++i) Gy[i * N + z] = y[i];
  }
  // metric terms are streamed one row (fixed y, all z) ahead; the empty asm statements keep the
  // compiler from hoisting every row's loads to the top (350 VGPRs of loads in flight)
  real_t mrow[2][N][8];
  STFEM_UNROLL
  for (int z = 0; z < N; ++z)
    STFEM_UNROLL
  for (int c = 0; c < 8; ++c) mrow[0][z][c] = met[(z * N * N + 0 * N + k) * 8 + c];
  STFEM_UNROLL
  for (int y = 0; y < N; ++y) {
    if (y + 1 < N) {
      // laundered offset: loads through it cannot be hoisted above this point (an offset, not the
      // pointer, so that the loads stay global_load: flat loads would also count as LDS traffic)
      int mo = 0;
      asm volatile("" : "+v"(mo));
      STFEM_UNROLL
      for (int z = 0; z < N; ++z)
        STFEM_UNROLL
      for (int c = 0; c < 8; ++c) mrow[(y + 1) & 1][z][c] = met[mo + (z * N * N + (y + 1) * N + k) * 8 + c];
    }
    real_t ur[N], gzr[N], fz[N], t[N];
    STFEM_UNROLL
    for (int z = 0; z < N; ++z) ur[z] = Ua[y * N + z];
    eo_apply<N, -1>(prm.eo_Dq, ur, gzr);
    STFEM_UNROLL
    for (int z = 0; z < N; ++z) {
      const int q = z * N * N + y * N + k;
      const real_t gx = cb_lds[q], gy = Gy[y * N + z], gz = gzr[z];
      const real_t *m = mrow[y & 1][z];
      const real_t fx = fma(m[0], gx, fma(m[1], gy, m[2] * gz));
      const real_t fy = fma(m[1], gx, fma(m[3], gy, m[4] * gz));
      fz[z] = fma(m[2], gx, fma(m[4], gy, m[5] * gz));
      R[y * N + z] *= m[6];
      Gy[y * N + z] = fy;
      if (out_active) cb_lds[q] = fx; // this lane's own column: read above, rewritten here
    }
    eo_apply<N, -1>(prm.eo_DqT, fz, t);
    STFEM_UNROLL
    for (int z = 0; z < N; ++z) R[y * N + z] += t[z];
    pin(R); // row y is complete before the next row's loads are issued
  }
  STFEM_UNROLL
  for (int z = 0; z < N; ++z) { // R += Dy^T Fy
    real_t x[N], y[N];
    STFEM_UNROLL
    for (int i = 0; i < N; ++i) x[i] = Gy[i * N + z];
    eo_apply<N, -1>(prm.eo_DqT, x, y);
    STFEM_UNROLL
    for (int i = 0; i < N; ++i) R[i * N + z] += y[i];
  }
  pin(R);
  wave_lds_fence();

  // ---- A4: Dx^T of the x flux in layout A
  STFEM_UNROLL
  for (int y = 0; y < N; ++y)
    STFEM_UNROLL
  for (int x = 0; x < N; ++x) PA[y * N + x] = cb_lds[k * N * N + y * N + x];
  plane_sweep<N, -1, true>(prm.eo_DqT, PA);
  wave_lds_fence();
  if (out_active) {
    STFEM_UNROLL
    for (int y = 0; y < N; ++y)
      STFEM_UNROLL
    for (int x = 0; x < N; ++x) cb_lds[k * N * N + y * N + x] = PA[y * N + x];
  }
  wave_lds_fence();

  // ---- B5: collect, integrate z
  STFEM_UNROLL
  for (int y = 0; y < N; ++y)
    STFEM_UNROLL
  for (int z = 0; z < N; ++z) R[y * N + z] += cb_lds[z * N * N + y * N + k];
  plane_sweep_T<N, true>(prm.eo_S, R);
  wave_lds_fence();
  if (out_active) {
    STFEM_UNROLL
    for (int y = 0; y < N; ++y)
      STFEM_UNROLL
    for (int z = 0; z < N; ++z) cb_lds[z * N * N + y * N + k] = R[y * N + z];
  }
  wave_lds_fence();

  // ---- A6: integrate y, x
  STFEM_UNROLL
  for (int y = 0; y < N; ++y)
    STFEM_UNROLL
  for (int x = 0; x < N; ++x) PA[y * N + x] = cb_lds[k * N * N + y * N + x];
  plane_sweep_T<N, false>(prm.eo_S, PA);
  plane_sweep_T<N, true>(prm.eo_S, PA);
  wave_lds_fence();
}

// "Pencil" form of cell_core (stfem_pencil.hip): the lanes of a cell-block run along the MEMORY x
// direction (lane = x-node i, registers [y][z]), so that the gather and the scatter of
// do_cell_integral_range (operators.h:1112-1133) are rows of contiguous doubles across the lanes
// and the z / y faces shared with the next cell of a marching wave stay in the lane's own
// registers.  The algorithm is the fast diagonalisation of cell_core with the roles of x and z
// exchanged; the three phases are separate functions so that the caller can issue the next
// cell group's src loads between them.
//   forward : nodal [y][z] -> modal in z, y -> LDS (plane of lane i)
//   middle  : lane k = z-mode; per y-mode row: read the x-lines of all input blocks, to modal
//             space in x, diagonal scaling + temporal combination, back to nodal values in x,
//             written straight back to the slab.  The x = P node of a cell IS the x = 0 node of
//             the next cell of the wave: its partial sum is added into the neighbour's slot
//             (ds_add_f64 after the neighbour's own write; LDS operations of one wave execute in
//             order), which resolves the x faces inside a wave without any shuffle.
//   backward: plane of lane i <- LDS, modal -> nodal in y, z.
template <int P, int NBM> struct PencilCore {
  using G = Geometry<P, NBM>;
  static constexpr int N = G::N, NN = N * N, CPW = G::CELLS_PER_WAVE;
  // Slab layout: cell-block (cell c, block b) at (b * CPW + c) * CBS, element (plane, row, col) at
  // plane * PS + row * N + col.  PS = 1 mod 16 doubles and CBS = N * PS make all four access
  // patterns of the core free of LDS bank conflicts for 8-byte elements (the lanes of a cell-block
  // step by PS doubles = 2 banks mod 32 in the plane-wise accesses and by one double in the
  // middle phase, consecutive cell-blocks continue where the previous one ends: tools/lds_conflicts.py);
  // with the dense strides 25 / 125 of Q4 nearly half of the LDS cycles were conflicts.
  static constexpr int PS = 16 * ((NN - 1 + 15) / 16) + 1;
  static constexpr int CBS = N * PS;
  static constexpr int LDS_PER_WAVE = CPW * NBM * CBS;
  static __device__ __forceinline__ int cb_offset(int cell, int blk) { return (blk * CPW + cell) * CBS; }

  static __device__ __forceinline__ void forward(const SweepParams &prm, real_t *__restrict__ cb_lds, int k,
                                                 bool in_active, real_t (&PA)[NN])
  {
    // (tables re-read from the kernel arguments at the start of every phase, see opaque_zero(): held
    // across the whole cell loop they cost ~50 SGPRs and push the loop's masks and offsets into VGPR lanes)
    const real_t *W = prm.fd_W + opaque_zero();
    fd_plane<N, true, true>(W, PA);
    fd_plane<N, true, false>(W, PA);
    if (in_active) {
      STFEM_UNROLL
      for (int y = 0; y < N; ++y)
        STFEM_UNROLL
      for (int x = 0; x < N; ++x) cb_lds[k * PS + y * N + x] = PA[y * N + x];
    }
    wave_lds_fence();
  }

  // is_last: this lane's cell is the last of the row and keeps its own x = P slot.
  // lzk: eigenvalue of this lane's z-mode (lane-constant: the caller keeps it; selecting it here
  // becomes an indexed global load from the kernel arguments inside the cell loop)
  // row_hook(y) runs before row y is computed: the caller spreads its vector-memory instructions
  // over the phase instead of issuing them in one burst (which fills the CU's memory pipeline and
  // stalls the wave at the issue of every further one).
  template <class Hook>
  static __device__ __forceinline__ void middle(const SweepParams &prm, real_t *__restrict__ lds, int cell_in_wave,
                                                int blk, int k, bool out_active, bool is_last, real_t lzk,
                                                const real_t (&aK)[NBM], const real_t (&aM)[NBM], Hook &&row_hook)
  {
    real_t *cb_lds = lds + cb_offset(cell_in_wave, blk);
    const int oz = opaque_zero();
    const real_t *W = prm.fd_W + oz, *lx = prm.fd_lx + oz, *ly = prm.fd_ly + oz;
    // opaque copies: see cell_core
    real_t wK[NBM], wM[NBM];
    STFEM_UNROLL
    for (int i = 0; i < NBM; ++i) {
      wK[i] = aK[i];
      wM[i] = aM[i];
      asm volatile("" : "+v"(wK[i]), "+v"(wM[i]));
    }
    asm volatile("" : "+v"(lzk));
    real_t vbuf[2][NBM][N];
    auto load_row = [&](int y, real_t (&v)[NBM][N]) {
      int base = cb_offset(cell_in_wave, 0) + y * N + k;
      asm volatile("" : "+v"(base));
      STFEM_UNROLL
      for (int i = 0; i < NBM; ++i)
        if (i == 0 || i < prm.nbi) {
          STFEM_UNROLL
          for (int x = 0; x < N; ++x) v[i][x] = lds[base + i * CPW * CBS + x * PS];
        }
    };
    load_row(0, vbuf[0]);
    STFEM_UNROLL
    for (int y = 0; y < N; ++y) {
      row_hook(y);
      if (y + 1 < N) load_row(y + 1, vbuf[(y + 1) & 1]);
      const real_t sy = lzk + ly[y];
      real_t acc[N];
      if (NBM <= 2) {
        STFEM_UNROLL
        for (int i = 0; i < NBM; ++i) {
          if (i == 0 || i < prm.nbi) {
            real_t t[N];
            fd_forward<N>(W, vbuf[y & 1][i], t);
            STFEM_UNROLL
            for (int x = 0; x < N; ++x) {
              const real_t d = fma(wK[i], sy + lx[x], wM[i]);
              acc[x] = i == 0 ? d * t[x] : fma(d, t[x], acc[x]);
            }
          }
        }
      } else {
        real_t ua[N], ub[N];
        STFEM_UNROLL
        for (int x = 0; x < N; ++x) ua[x] = ub[x] = real_t(0);
        STFEM_UNROLL
        for (int i = 0; i < NBM; ++i) {
          if (i < prm.nbi) {
            STFEM_UNROLL
            for (int x = 0; x < N; ++x) {
              ua[x] = fma(wK[i], vbuf[y & 1][i][x], ua[x]);
              ub[x] = fma(wM[i], vbuf[y & 1][i][x], ub[x]);
            }
          }
        }
        real_t ta[N], tb[N];
        fd_forward<N>(W, ua, ta);
        fd_forward<N>(W, ub, tb);
        STFEM_UNROLL
        for (int x = 0; x < N; ++x) acc[x] = fma(sy + lx[x], ta[x], tb[x]);
      }
      real_t r[N];
      fd_backward<N>(W, acc, r);
      pin(r);
      if (out_active) {
        STFEM_UNROLL
        for (int x = 0; x < P; ++x) cb_lds[x * PS + y * N + k] = r[x];
        asm volatile("" ::: "memory"); // the neighbour's own x = 0 write precedes the add below
        if (is_last) cb_lds[P * PS + y * N + k] = r[P];
        else atomicAdd(&cb_lds[CBS + y * N + k], r[P]); // the next cell of the row, same block
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    wave_lds_fence();
  }

  // The middle phase for systems of four and more temporal blocks: the input blocks' x-lines are STREAMED (one block
  // ahead) into the two weighted sums ua = sum_i aK(j,i) u_i, ub = sum_i aM(j,i) u_i instead of being held all at once
  // (2 x NBM x N registers), and the weights of this lane's output block come from a small LDS table, wrow[2 i] =
  // aK(j,i) vol, wrow[2 i + 1] = aM(j,i) vol (2 x NBM registers per lane otherwise, for the whole kernel).  The
  // cell-wise coefficients fK, fM (operators.h:1060-1087) multiply the two sums after the transform.
  template <class Hook>
  static __device__ __forceinline__ void middle_stream(const SweepParams &prm, real_t *__restrict__ lds, int cell_in_wave,
                                                       int blk, int k, bool out_active, bool is_last, real_t lzk,
                                                       const real_t *__restrict__ wrow, real_t fK, real_t fM, Hook &&row_hook)
  {
    real_t *cb_lds = lds + cb_offset(cell_in_wave, blk);
    const int oz = opaque_zero();
    const real_t *W = prm.fd_W + oz, *lx = prm.fd_lx + oz, *ly = prm.fd_ly + oz;
    const int nbi = prm.nbi;
    asm volatile("" : "+v"(lzk), "+v"(fK), "+v"(fM));
    STFEM_UNROLL
    for (int y = 0; y < N; ++y) {
      row_hook(y);
      int base = cb_offset(cell_in_wave, 0) + y * N + k;
      asm volatile("" : "+v"(base));
      real_t ua[N], ub[N], v[2][N], w[2][2];
      STFEM_UNROLL
      for (int x = 0; x < N; ++x) {
        ua[x] = ub[x] = real_t(0);
        v[0][x] = lds[base + x * PS];
      }
      w[0][0] = wrow[0];
      w[0][1] = wrow[1];
      STFEM_UNROLL
      for (int i = 0; i < NBM; ++i) {
        if (i < nbi) { // (wave-uniform)
          if (i + 1 < NBM && i + 1 < nbi) {
            STFEM_UNROLL
            for (int x = 0; x < N; ++x) v[(i + 1) & 1][x] = lds[base + (i + 1) * CPW * CBS + x * PS];
            w[(i + 1) & 1][0] = wrow[2 * (i + 1)];
            w[(i + 1) & 1][1] = wrow[2 * (i + 1) + 1];
          }
          STFEM_UNROLL
          for (int x = 0; x < N; ++x) {
            ua[x] = fma(w[i & 1][0], v[i & 1][x], ua[x]);
            ub[x] = fma(w[i & 1][1], v[i & 1][x], ub[x]);
          }
        }
      }
      const real_t sy = lzk + ly[y];
      real_t ta[N], tb[N], acc[N], r[N];
      fd_forward<N>(W, ua, ta);
      fd_forward<N>(W, ub, tb);
      STFEM_UNROLL
      for (int x = 0; x < N; ++x) acc[x] = fma(fK * (sy + lx[x]), ta[x], fM * tb[x]);
      fd_backward<N>(W, acc, r);
      pin(r);
      if (out_active) {
        STFEM_UNROLL
        for (int x = 0; x < P; ++x) cb_lds[x * PS + y * N + k] = r[x];
        asm volatile("" ::: "memory"); // the neighbour's own x = 0 write precedes the add below
        if (is_last) cb_lds[P * PS + y * N + k] = r[P];
        else atomicAdd(&cb_lds[CBS + y * N + k], r[P]); // the next cell of the row, same block
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    wave_lds_fence();
  }

  // plane of lane i <- LDS, modal -> nodal in y (all columns), then row by row in z;
  // row_done(y, r) receives the finished row y of the result plane, r[z], as soon as it is complete
  template <class RowDone>
  static __device__ __forceinline__ void backward(const SweepParams &prm, const real_t *__restrict__ cb_lds, int k,
                                                  RowDone &&row_done)
  {
    const real_t *W = prm.fd_W + opaque_zero();
    real_t R[NN];
    STFEM_UNROLL
    for (int y = 0; y < N; ++y)
      STFEM_UNROLL
    for (int x = 0; x < N; ++x) R[y * N + x] = cb_lds[k * PS + y * N + x];
    fd_plane<N, false, false>(W, R);
    wave_lds_fence();
    STFEM_UNROLL
    for (int y = 0; y < N; ++y) {
      real_t x[N], r[N];
      STFEM_UNROLL
      for (int z = 0; z < N; ++z) x[z] = R[y * N + z];
      fd_backward<N>(W, x, r);
      row_done(y, r);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
};

// Dirichlet flags of the plane (cell, k): which local rows/columns are constrained.
struct PlaneMask {
  bool x0, x1, y0, y1, all;
};
template <int P>
__device__ __forceinline__ PlaneMask plane_mask(const SweepParams &prm, int cx, int cy, int cz, int k)
{
  PlaneMask m;
  m.x0 = (prm.dmask & 1) && cx == 0;
  m.x1 = (prm.dmask & 2) && cx == prm.ncx - 1;
  m.y0 = (prm.dmask & 4) && cy == 0;
  m.y1 = (prm.dmask & 8) && cy == prm.ncy - 1;
  m.all = ((prm.dmask & 16) && cz == 0 && k == 0) || ((prm.dmask & 32) && cz == prm.ncz - 1 && k == P);
  return m;
}
template <int P> __device__ __forceinline__ bool constrained(const PlaneMask &m, int y, int x)
{
  return m.all || (x == 0 && m.x0) || (x == P && m.x1) || (y == 0 && m.y0) || (y == P && m.y1);
}

constexpr int round_nbm(int nbm) { return nbm <= 4 ? nbm : (nbm <= 6 ? 6 : 8); }

} // namespace STFEM_PREC
} // namespace stfem
