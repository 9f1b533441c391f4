// The class-block apply kernel of the cell-patch smoothers (stfem_vanka.hip: scalar systems; stfem_stokes_vanka.hip: the
// two-variable Stokes system), shared by the two translation units.  Not part of the boundary.
#pragma once
#include <hip/hip_runtime.h>

namespace {

constexpr int VK_MAX_BLOCKS = 8;
constexpr int VK_MAX_ROWS = 512; // rows of a cell block (16 x 32 tiles: Q4 with four temporal blocks = cG(2), two time steps per slab)
constexpr long long VK_NO_ROW = -0x7fffffffffffffffll - 1; // offset-table entry of a row beyond the block (pointer differences may be negative)
constexpr int KS = 16; // k rows of the inverse staged in LDS per step (two buffers)

struct VankaParams {
  const void *src[VK_MAX_BLOCKS];
  void *dst[VK_MAX_BLOCKS];
  const void *blocks; // [class][kpad][mpad], element (row r, column k) of the inverse at [k][r]
  const int *off;     // local node -> DoF offset from the cell's first node
  const int *cell;    // [nquad * 64]: first DoF of the cell, -1 = padding
  const int *cls;     // [nquad]: class index | neighbour pattern << 8 (2 bits per direction: has lower, has upper)
  int nquad, m, mpad, kpad;
  int colour;         // (cx & 1) + 2 (cy & 1) + 4 (cz & 1) of the cells of this launch
  int p;
  void *flat;         // two-phase apply (small meshes): Y[slot][mpad], slot = 64 quad + 16 wave + column; nullptr: colour launches
  double omega;       // dst = (accumulate ? dst : 0) + omega * (sum over cells ...)  (the relaxation step around the smoother)
  int accumulate;
  // two-variable systems (Stokes; two-phase apply only): row r reads element rowtab[r].y of vector rowtab[r].x & 255, counted
  // from the cell's first DoF of variable rowtab[r].x >> 8 (0: cell[], 1: cell2[])
  const int2 *rowtab;
  const int *cell2;
};

template <typename T> struct Mfma;
template <> struct Mfma<double> {
  typedef double acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  // C/D of v_mfma_f64_16x16x4_f64: column = lane & 15, row = (lane >> 4) + 4 * reg
  static __device__ __forceinline__ int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <> struct Mfma<float> {
  typedef float acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  // C/D of v_mfma_f32_16x16x4_f32: column = lane & 15, row = 4 * (lane >> 4) + reg
  static __device__ __forceinline__ int row(int lane, int reg) { return 4 * (lane >> 4) + reg; }
};

// One workgroup = four waves = four batches of 16 cells of ONE block class; a wave holds the 16 x 16
// accumulator tiles of all MT row tiles of its 16 cells.  The rows of the inverse pass through LDS in
// slabs of KS, shared by the four waves.
template <typename T, int NLOC, int MT>
__global__ __launch_bounds__(256, 2) void vanka_apply_kernel(const VankaParams prm)
{
  using M = Mfma<T>;
  constexpr int MPAD = 16 * MT;               // rows this workgroup computes: [blockIdx.y MPAD, (blockIdx.y + 1) MPAD)
  __shared__ T slab[2][KS * MPAD];
  // byte offset of row r = (block, local node) of X / Y from the first source / destination block, for the cell
  // whose first DoF is 0; VK_NO_ROW beyond the last row.  (Indexing the kernel arguments with a lane's block number would
  // make every gather a dependent pair of global loads.)  Bit 0 of a destination entry: this cell is the first of
  // the eight colour launches to touch the DoF - it stores, the later ones add (see stfem_vanka_vmult).
  __shared__ long long s_src[VK_MAX_ROWS], s_dst[MPAD];
  const int row0 = blockIdx.y * MPAD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int quad = blockIdx.x;
  const int cls = prm.cls[quad] & 255, pattern = prm.cls[quad] >> 8;
  for (int r = threadIdx.x; r < VK_MAX_ROWS; r += 256) {
    long long os = VK_NO_ROW, od = VK_NO_ROW;
    if (r < prm.m) {
      int blk = r / NLOC, n = r - blk * NLOC, var = 0;
      long long o;
      if (prm.rowtab) {
        const int2 e = prm.rowtab[r];
        blk = e.x & 255; var = e.x >> 8; n = 0;
        o = (long long)e.y * (long long)sizeof(T);
      } else
        o = (long long)prm.off[n] * (long long)sizeof(T);
#pragma unroll
      for (int b = 0; b < VK_MAX_BLOCKS; ++b)
        if (b == blk) {
          os = (static_cast<const char *>(prm.src[b]) - static_cast<const char *>(prm.src[0])) + o;
          od = (static_cast<char *>(prm.dst[b]) - static_cast<char *>(prm.dst[0])) + o;
        }
      // a DoF on a face shared with a neighbour is first touched by the cell whose colour bit is 0 in every
      // shared direction (the launches run in ascending colour order)
      const int np = prm.p + 1;
      const int idx[3] = {n % np, (n / np) % np, n / (np * np)};
      bool first = true;
#pragma unroll
      for (int dd = 0; dd < 3; ++dd) {
        const int k = (pattern >> (2 * dd)) & 3;
        const bool shared = (idx[dd] == 0 && (k & 1)) || (idx[dd] == prm.p && (k & 2));
        if (shared && ((prm.colour >> dd) & 1)) first = false;
      }
      if (first && !prm.accumulate) od |= 1;
      os |= (long long)(var << 1); // (offsets are multiples of sizeof(T) >= 4: bits 0 and 1 are free)
    }
    s_src[r] = os;
    if (r >= row0 && r < row0 + MPAD) s_dst[r - row0] = od;
  }
  // padded row tiles of the last part may lie beyond the table (parts * MPAD > VK_MAX_ROWS): no rows there
  for (int r = threadIdx.x; r < MPAD; r += 256)
    if (row0 + r >= VK_MAX_ROWS) s_dst[r] = VK_NO_ROW;
  // the inverse of this class, [kpad][mpad] with mpad = all row tiles; this workgroup's MPAD columns of every k row
  const T *Binv = static_cast<const T *>(prm.blocks) + size_t(cls) * prm.kpad * prm.mpad + row0;
  const int base = prm.cell[(quad * 4 + wave) * 16 + (lane & 15)]; // this lane's cell (column of X and Y)
  const char *src0 = static_cast<const char *>(prm.src[0]) + (long long)(base < 0 ? 0 : base) * (long long)sizeof(T);
  const char *src1 = src0;
  if (prm.cell2) src1 = static_cast<const char *>(prm.src[0]) + (long long)max(prm.cell2[(quad * 4 + wave) * 16 + (lane & 15)], 0) * (long long)sizeof(T);
  char *dst0 = static_cast<char *>(prm.dst[0]) + (long long)(base < 0 ? 0 : base) * (long long)sizeof(T);
  // rows [s KS, (s + 1) KS) of the (padded) inverse: fetched into registers DEPTH slabs ahead of their use, written to the
  // other LDS buffer after the slab before has been multiplied.  (One slab ahead, as in round 2, leaves a full global-load
  // latency in every step of the k loop: on the small multigrid levels, where a launch is a handful of workgroups, a step took
  // 1.7 us.  global_load_lds_dwordx4 straight into LDS - no registers, no ds_write - measured 3-10 % SLOWER: profiles/r2/vanka.)
  constexpr int DEPTH = 3;
  constexpr int SR = (KS * MPAD + 255) / 256; // slab elements every thread moves
  T sreg[DEPTH][SR];
  auto fetch = [&](int s, T (&reg)[SR]) {
    const T *g = Binv + size_t(s) * KS * prm.mpad;
#pragma unroll
    for (int q = 0; q < SR; ++q) {
      const int e = q * 256 + int(threadIdx.x); // element (k row e / MPAD, column e % MPAD) of the slab
      if (KS * MPAD % 256 == 0 || e < KS * MPAD) reg[q] = g[(e / MPAD) * prm.mpad + e % MPAD];
    }
  };
  auto deposit = [&](int buf, const T (&reg)[SR]) {
#pragma unroll
    for (int q = 0; q < SR; ++q)
      if (KS * MPAD % 256 == 0 || q * 256 + int(threadIdx.x) < KS * MPAD) slab[buf][q * 256 + threadIdx.x] = reg[q];
  };
  typename M::acc_t acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = typename M::acc_t{0, 0, 0, 0};
  // this lane's row of X in a k-step: krow = 4 step + (lane >> 4); the values are gathered DEPTH slabs ahead, in slab order
  int krow = lane >> 4;
  auto gather = [&]() -> T {
    T v = T(0);
    if (krow < VK_MAX_ROWS) {
      const long long o = s_src[krow];
      if (base >= 0 && o != VK_NO_ROW) v = *reinterpret_cast<const T *>(((o & 2) ? src1 : src0) + (o & ~3ll));
    }
    krow += 4;
    return v;
  };
  const int nslab = prm.kpad / KS;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (d < nslab) fetch(d, sreg[d]);
  __syncthreads(); // the offset tables
  T xr[DEPTH][KS / 4];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (d < nslab) {
#pragma unroll
      for (int q = 0; q < KS / 4; ++q) xr[d][q] = gather();
    }
  deposit(0, sreg[0]);
  __syncthreads();
  for (int s0 = 0; s0 < nslab; s0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { // slab s lives in register slot s % DEPTH = d
      const int s = s0 + d;
      if (s < nslab) {                // (uniform)
        const bool more = s + DEPTH < nslab;
        T xnew[KS / 4];
        if (more) {
          fetch(s + DEPTH, sreg[d]);  // slot d went to LDS in the step before
#pragma unroll
          for (int q = 0; q < KS / 4; ++q) xnew[q] = gather();
        }
        const T *sl = slab[s & 1];
#pragma unroll
        for (int q = 0; q < KS / 4; ++q) {
          const T *a = sl + (4 * q + (lane >> 4)) * MPAD + (lane & 15);
#pragma unroll
          for (int t = 0; t < MT; ++t) acc[t] = M::mma(a[16 * t], xr[d][q], acc[t]);
        }
        if (more) {
#pragma unroll
          for (int q = 0; q < KS / 4; ++q) xr[d][q] = xnew[q];
        }
        if (s + 1 < nslab) deposit((s + 1) & 1, sreg[(d + 1) % DEPTH]);
        __syncthreads();
      }
    }
  }
  if (prm.flat) { // two-phase apply: the cell's rows go to the scratch array, vanka_collect_kernel sums them per DoF
    T *y = static_cast<T *>(prm.flat) + (size_t(quad) * 64 + wave * 16 + (lane & 15)) * prm.mpad + row0;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) y[16 * t + M::row(lane, r)] = acc[t][r];
    return;
  }
  // scatter: rows of Y back to the DoFs of the cell (cells of one launch share none).  All loads first
  // (first touches, rows beyond the block and padding cells load nothing), then the stores.
  // (two row tiles = eight loads in flight per lane at a time: more only costs registers)
  constexpr int TC = MT >= 2 ? 2 : 1;
  const T om = T(prm.omega);
#pragma unroll
  for (int t0 = 0; t0 < MT; t0 += TC) {
    T *d[TC * 4];
    T old[TC * 4];
#pragma unroll
    for (int t = 0; t < TC; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long o = t0 + t < MT ? s_dst[16 * (t0 + t) + M::row(lane, r)] : VK_NO_ROW;
        d[4 * t + r] = (base >= 0 && o != VK_NO_ROW) ? reinterpret_cast<T *>(dst0 + (o & ~1ll)) : nullptr;
        old[4 * t + r] = (d[4 * t + r] && !(o & 1)) ? *d[4 * t + r] : T(0);
      }
#pragma unroll
    for (int t = 0; t < TC; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (t0 + t < MT && d[4 * t + r]) *d[4 * t + r] = old[4 * t + r] + om * acc[t0 + t][r];
  }
}

} // namespace
