// Device-side building blocks of the fused space-time cell sweep (gfx950, wave64).
//
// Thread layout ("plane per thread"): a cell's (p+1)^3 tensor of one temporal block is held by
// N = p+1 lanes of ONE wave, each owning one N x N plane in registers.  Two of the three 1D
// contractions of every sum-factorisation stage run entirely in registers; the third direction
// is reached by a transpose through a wave-private LDS slab (no workgroup barrier: LDS
// operations of one wave execute in order).
//
//   layout A: lane index = z-plane,  registers [y][x]
//   layout B: lane index = x-plane,  registers [y][z]
//
// 1D matrices are applied in even-odd form (Kopriva / Kronbichler-Kormann): 21 instead of 25
// multiply-adds for N = 5, and only eo_size(N) distinct constants, which live in SGPRs.
#pragma once
#include "stfem_kernels.h"

#include <hip/hip_runtime.h>

namespace stfem {
namespace STFEM_PREC {

#define STFEM_UNROLL _Pragma("unroll")

__host__ __device__ constexpr int eo_size_c(int n)
{
  return 2 * (n / 2) * (n / 2) + ((n & 1) ? 2 * (n / 2) + 1 : 0);
}

// y = X x for an N x N matrix with X[q][a] = SIGN * X[N-1-q][N-1-a], constants packed by
// host_tables.cpp::eo_pack.  `c` must be wave-uniform (kernel argument -> SGPRs).
template <int N, int SIGN>
__device__ __forceinline__ void eo_apply(const real_t *__restrict__ c, const real_t (&x)[N],
                                         real_t (&y)[N])
{
  constexpr int H = N / 2;
  constexpr bool ODD = (N & 1) != 0;
  real_t xe[H > 0 ? H : 1], xo[H > 0 ? H : 1];
  STFEM_UNROLL
  for (int i = 0; i < H; ++i) {
    xe[i] = x[i] + x[N - 1 - i];
    xo[i] = x[i] - x[N - 1 - i];
  }
  STFEM_UNROLL
  for (int q = 0; q < H; ++q) {
    real_t a = c[q * H] * xe[0];
    real_t b = c[H * H + q * H] * xo[0];
    STFEM_UNROLL
    for (int i = 1; i < H; ++i) {
      a = fma(c[q * H + i], xe[i], a);
      b = fma(c[H * H + q * H + i], xo[i], b);
    }
    if (ODD) a = fma(c[2 * H * H + q], x[H], a);
    y[q] = a + b;
    y[N - 1 - q] = SIGN > 0 ? a - b : b - a;
  }
  if (ODD) {
    real_t m;
    if (SIGN > 0) {
      m = c[2 * H * H + 2 * H] * x[H];
      STFEM_UNROLL
      for (int i = 0; i < H; ++i) m = fma(c[2 * H * H + H + i], xe[i], m);
    } else {
      m = c[2 * H * H + H] * xo[0];
      STFEM_UNROLL
      for (int i = 1; i < H; ++i) m = fma(c[2 * H * H + H + i], xo[i], m);
    }
    y[H] = m;
  }
}

// y = X^T x for a symmetric-type X packed in c (the transpose has the same constants:
// ee^T[q][i] = ee[i][q], oo^T[q][i] = oo[i][q], em <-> mm), so S and S^T share their SGPRs.
template <int N>
__device__ __forceinline__ void eo_apply_T(const real_t *__restrict__ c, const real_t (&x)[N],
                                           real_t (&y)[N])
{
  constexpr int H = N / 2;
  constexpr bool ODD = (N & 1) != 0;
  real_t xe[H > 0 ? H : 1], xo[H > 0 ? H : 1];
  STFEM_UNROLL
  for (int i = 0; i < H; ++i) {
    xe[i] = x[i] + x[N - 1 - i];
    xo[i] = x[i] - x[N - 1 - i];
  }
  STFEM_UNROLL
  for (int q = 0; q < H; ++q) {
    real_t a = c[q] * xe[0];
    real_t b = c[H * H + q] * xo[0];
    STFEM_UNROLL
    for (int i = 1; i < H; ++i) {
      a = fma(c[i * H + q], xe[i], a);
      b = fma(c[H * H + i * H + q], xo[i], b);
    }
    if (ODD) a = fma(c[2 * H * H + H + q], x[H], a); // em^T = mm
    y[q] = a + b;
    y[N - 1 - q] = a - b;
  }
  if (ODD) {
    real_t m = c[2 * H * H + 2 * H] * x[H];
    STFEM_UNROLL
    for (int i = 0; i < H; ++i) m = fma(c[2 * H * H + i], xe[i], m); // mm^T = em
    y[H] = m;
  }
}

// In-place sweep with the transposed matrix
template <int N, bool ALONG_FAST>
__device__ __forceinline__ void plane_sweep_T(const real_t *__restrict__ c, real_t (&P)[N * N])
{
  STFEM_UNROLL
  for (int o = 0; o < N; ++o) {
    real_t x[N], y[N];
    STFEM_UNROLL
    for (int i = 0; i < N; ++i) x[i] = ALONG_FAST ? P[o * N + i] : P[i * N + o];
    eo_apply_T<N>(c, x, y);
    STFEM_UNROLL
    for (int i = 0; i < N; ++i) (ALONG_FAST ? P[o * N + i] : P[i * N + o]) = y[i];
  }
}

// In-place sweep over an N x N register plane P[s*N + f] (s slow, f fast index).
// ALONG_FAST: contract the fast index for every slow index; else contract the slow index.
template <int N, int SIGN, bool ALONG_FAST>
__device__ __forceinline__ void plane_sweep(const real_t *__restrict__ c, real_t (&P)[N * N])
{
  STFEM_UNROLL
  for (int o = 0; o < N; ++o) {
    real_t x[N], y[N];
    STFEM_UNROLL
    for (int i = 0; i < N; ++i) x[i] = ALONG_FAST ? P[o * N + i] : P[i * N + o];
    eo_apply<N, SIGN>(c, x, y);
    STFEM_UNROLL
    for (int i = 0; i < N; ++i) (ALONG_FAST ? P[o * N + i] : P[i * N + o]) = y[i];
  }
}

// R += s * D^T ( D U ) along one in-register direction (collocation Laplacian of that
// direction with the quadrature weights folded into D).  D antisymmetric-type, as is D^T.
template <int N, bool ALONG_FAST>
__device__ __forceinline__ void plane_laplace_acc(const real_t *__restrict__ cD,
                                                  const real_t *__restrict__ cDT, real_t s,
                                                  const real_t (&U)[N * N], real_t (&R)[N * N])
{
  STFEM_UNROLL
  for (int o = 0; o < N; ++o) {
    real_t x[N], t[N], y[N];
    STFEM_UNROLL
    for (int i = 0; i < N; ++i) x[i] = ALONG_FAST ? U[o * N + i] : U[i * N + o];
    eo_apply<N, -1>(cD, x, t);
    eo_apply<N, -1>(cDT, t, y);
    STFEM_UNROLL
    for (int i = 0; i < N; ++i) {
      real_t &r = ALONG_FAST ? R[o * N + i] : R[i * N + o];
      r = fma(s, y[i], r);
    }
  }
}

// Same, overwriting U in place with s * D^T D U (used for the direction that needs its own
// layout, whose result is then transposed and added).
template <int N, bool ALONG_FAST>
__device__ __forceinline__ void plane_laplace_inplace(const real_t *__restrict__ cD,
                                                      const real_t *__restrict__ cDT, real_t s,
                                                      real_t (&U)[N * N])
{
  STFEM_UNROLL
  for (int o = 0; o < N; ++o) {
    real_t x[N], t[N], y[N];
    STFEM_UNROLL
    for (int i = 0; i < N; ++i) x[i] = ALONG_FAST ? U[o * N + i] : U[i * N + o];
    eo_apply<N, -1>(cD, x, t);
    eo_apply<N, -1>(cDT, t, y);
    STFEM_UNROLL
    for (int i = 0; i < N; ++i) (ALONG_FAST ? U[o * N + i] : U[i * N + o]) = s * y[i];
  }
}

// R += s * (X U) along one in-register direction (X of symmetric type)
template <int N, bool ALONG_FAST>
__device__ __forceinline__ void plane_sweep_acc(const real_t *__restrict__ c, real_t s,
                                                const real_t (&U)[N * N], real_t (&R)[N * N])
{
  STFEM_UNROLL
  for (int o = 0; o < N; ++o) {
    real_t x[N], y[N];
    STFEM_UNROLL
    for (int i = 0; i < N; ++i) x[i] = ALONG_FAST ? U[o * N + i] : U[i * N + o];
    eo_apply<N, +1>(c, x, y);
    STFEM_UNROLL
    for (int i = 0; i < N; ++i) {
      real_t &r = ALONG_FAST ? R[o * N + i] : R[i * N + o];
      r = fma(s, y[i], r);
    }
  }
}

// P <- s * (X P) in place
template <int N, bool ALONG_FAST>
__device__ __forceinline__ void plane_sweep_scaled(const real_t *__restrict__ c, real_t s,
                                                   real_t (&P)[N * N])
{
  STFEM_UNROLL
  for (int o = 0; o < N; ++o) {
    real_t x[N], y[N];
    STFEM_UNROLL
    for (int i = 0; i < N; ++i) x[i] = ALONG_FAST ? P[o * N + i] : P[i * N + o];
    eo_apply<N, +1>(c, x, y);
    STFEM_UNROLL
    for (int i = 0; i < N; ++i) (ALONG_FAST ? P[o * N + i] : P[i * N + o]) = s * y[i];
  }
}

// ---- fast diagonalisation (Cartesian cells, coefficient constant in the cell) ----
// Modes of the 1D pair (M1, K1) are even or odd (host_tables.h: fd_W), so the transform to
// modal space needs no recombination of its outputs and the transform back none of its inputs:
// 4 adds + 13 multiply-adds per line for N = 5, against 8 + 13 of a general even-odd product.
// y[0..NE) even modes, y[NE..N) odd modes.
template <int N>
__device__ __forceinline__ void fd_forward(const real_t *__restrict__ c, const real_t (&x)[N],
                                           real_t (&y)[N])
{
  constexpr int H = N / 2, NE = N - H;
  real_t xe[NE], xo[H > 0 ? H : 1];
  STFEM_UNROLL
  for (int i = 0; i < H; ++i) {
    xe[i] = x[i] + x[N - 1 - i];
    xo[i] = x[i] - x[N - 1 - i];
  }
  if (N & 1) xe[H] = x[H];
  STFEM_UNROLL
  for (int m = 0; m < NE; ++m) {
    real_t a = c[m * NE] * xe[0];
    STFEM_UNROLL
    for (int i = 1; i < NE; ++i) a = fma(c[m * NE + i], xe[i], a);
    y[m] = a;
  }
  STFEM_UNROLL
  for (int m = 0; m < H; ++m) {
    real_t b = c[NE * NE + m * H] * xo[0];
    STFEM_UNROLL
    for (int i = 1; i < H; ++i) b = fma(c[NE * NE + m * H + i], xo[i], b);
    y[NE + m] = b;
  }
}

// y = W^T x: modal -> nodal
template <int N>
__device__ __forceinline__ void fd_backward(const real_t *__restrict__ c, const real_t (&x)[N],
                                            real_t (&y)[N])
{
  constexpr int H = N / 2, NE = N - H;
  STFEM_UNROLL
  for (int i = 0; i < H; ++i) {
    real_t a = c[i] * x[0];
    STFEM_UNROLL
    for (int m = 1; m < NE; ++m) a = fma(c[m * NE + i], x[m], a);
    real_t b = c[NE * NE + i] * x[NE];
    STFEM_UNROLL
    for (int m = 1; m < H; ++m) b = fma(c[NE * NE + m * H + i], x[NE + m], b);
    y[i] = a + b;
    y[N - 1 - i] = a - b;
  }
  if (N & 1) {
    real_t a = c[H] * x[0];
    STFEM_UNROLL
    for (int m = 1; m < NE; ++m) a = fma(c[m * NE + H], x[m], a);
    y[H] = a;
  }
}

template <int N, bool FORWARD, bool ALONG_FAST>
__device__ __forceinline__ void fd_plane(const real_t *__restrict__ c, real_t (&P)[N * N])
{
  STFEM_UNROLL
  for (int o = 0; o < N; ++o) {
    real_t x[N], y[N];
    STFEM_UNROLL
    for (int i = 0; i < N; ++i) x[i] = ALONG_FAST ? P[o * N + i] : P[i * N + o];
    if (FORWARD) fd_forward<N>(c, x, y);
    else fd_backward<N>(c, x, y);
    STFEM_UNROLL
    for (int i = 0; i < N; ++i) (ALONG_FAST ? P[o * N + i] : P[i * N + o]) = y[i];
  }
}

// Orders the LDS traffic of ONE wave: everything this wave wrote before is visible to its
// later reads.  No instruction is emitted beyond what the compiler needs for its own ordering.
__device__ __forceinline__ void wave_lds_fence()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // also a scheduling barrier: without it the machine scheduler interleaves all five phases to
  // hide LDS latency and needs > 256 VGPRs (1 wave/SIMD); phase-ordered code needs ~half
  __builtin_amdgcn_sched_barrier(0);
}

// A zero the compiler cannot see through.  `table + opaque_zero()` makes the scalar loads of a 1D
// table (kernel argument -> s_load) depend on this program point, so they are re-issued where
// they are used instead of being hoisted out of the layer loop and held in ~50 SGPRs for the whole
// kernel - which pushes every lane mask and pointer into VGPR lanes (v_readlane per use).
__device__ __forceinline__ int opaque_zero()
{
  int z = 0;
  asm volatile("" : "+s"(z));
  return z;
}

// Pins a register plane at this program point: everything that produces it is scheduled before,
// everything that consumes it after.  (The DAG scheduler otherwise sinks whole sweeps next to
// their far-away use and keeps three extra planes alive.)  Emits no instruction.
template <int M> __device__ __forceinline__ void pin(real_t (&P)[M])
{
  STFEM_UNROLL
  for (int e = 0; e < M; ++e) asm volatile("" : "+v"(P[e]));
}

} // namespace STFEM_PREC
} // namespace stfem
