// Fused space-time cell sweep for CDNA4 (gfx950).
//
// Computes, for every cell and all temporal blocks at once,
//     dst_j += sum_i alpha(j,i) K_cell src_i + beta(j,i) M_cell src_i
// which is the per-cell body of SystemMatrix::vmult (reference include/operators.h:536-559)
// around MatrixFreeOperator::do_cell_integral_local (operators.h:1135-1173), restructured:
// the temporal combination commutes with the spatial interpolation, so it is applied once to
// the x/y-interpolated data and the K and M parts share one evaluate/integrate pipeline
// (10 one-dimensional sweeps per output block instead of 2 x 12 per input block).
#include "stfem_device.h"
#include "stfem_kernels.h"

#include <hip/hip_runtime.h>

#include <cstdlib>

namespace stfem {

namespace {

template <int P, int NBM> struct Geometry {
  static constexpr int N = P + 1;
  static constexpr int CB_PER_WAVE = 64 / N;            // cell-blocks (cell x temporal block) per wave
  static constexpr int CELLS_PER_WAVE = CB_PER_WAVE / NBM;
  static constexpr int ACTIVE = CELLS_PER_WAVE * NBM * N; // active lanes
  static constexpr int CBS = N * N * N;                   // LDS doubles per cell-block
  static constexpr int WAVES = 4;
  static constexpr int LDS_PER_WAVE = CELLS_PER_WAVE * NBM * CBS;
};

// One pass of the fused operator over the cells owned by this wave.
// PA: on entry the nodal src plane (layout A, [y][x]) of (cell, input block blk, z-plane k),
//     on exit the nodal result plane of (cell, output block blk, z-plane k).
template <int P, int NBM>
__device__ __forceinline__ void
cell_core(const SweepParams &prm, double *__restrict__ lds, int cell_in_wave, int blk, int k,
          bool in_active, bool out_active, const double (&aK)[NBM], const double (&aM)[NBM],
          double (&PA)[(P + 1) * (P + 1)])
{
  using G = Geometry<P, NBM>;
  constexpr int N = G::N;
  constexpr int CBS = G::CBS;
  double *cb_lds = lds + (cell_in_wave * NBM + blk) * CBS;

  // ---- phase A: interpolate x, y (registers), hand over to layout B
  plane_sweep<N, +1, true>(prm.eo_Si, PA);
  plane_sweep<N, +1, false>(prm.eo_Si, PA);
  if (in_active) {
    STFEM_UNROLL
    for (int y = 0; y < N; ++y)
      STFEM_UNROLL
    for (int x = 0; x < N; ++x) cb_lds[k * N * N + y * N + x] = PA[y * N + x];
  }
  wave_lds_fence();

  // ---- phase B: temporal combination, interpolate z, mass + y/z Laplacian
  double Ua[N * N], R[N * N];
  STFEM_UNROLL
  for (int e = 0; e < N * N; ++e) Ua[e] = R[e] = 0.0;
  STFEM_UNROLL
  for (int i = 0; i < NBM; ++i) {
    if (i < prm.nbi) {
      const double *in_lds = lds + (cell_in_wave * NBM + i) * CBS;
      STFEM_UNROLL
      for (int y = 0; y < N; ++y)
        STFEM_UNROLL
      for (int z = 0; z < N; ++z) {
        const double v = in_lds[z * N * N + y * N + k];
        Ua[y * N + z] = fma(aK[i], v, Ua[y * N + z]);
        R[y * N + z] = fma(aM[i], v, R[y * N + z]);
      }
    }
  }
  plane_sweep<N, +1, true>(prm.eo_Si, Ua);
  plane_sweep<N, +1, true>(prm.eo_Si, R);
  // Cartesian cell, coefficient constant in the cell: D^T c D collapses to c * L (one sweep)
  plane_sweep_acc<N, true>(prm.eo_L, prm.ihz2, Ua, R);
  plane_sweep_acc<N, false>(prm.eo_L, prm.ihy2, Ua, R);
  pin(R);
  wave_lds_fence();
  if (out_active) {
    STFEM_UNROLL
    for (int y = 0; y < N; ++y)
      STFEM_UNROLL
    for (int z = 0; z < N; ++z) cb_lds[z * N * N + y * N + k] = Ua[y * N + z];
  }
  wave_lds_fence();

  // ---- phase A2: x Laplacian in layout A
  STFEM_UNROLL
  for (int y = 0; y < N; ++y)
    STFEM_UNROLL
  for (int x = 0; x < N; ++x) PA[y * N + x] = cb_lds[k * N * N + y * N + x];
  plane_sweep_scaled<N, true>(prm.eo_L, prm.ihx2, PA);
  wave_lds_fence();
  if (out_active) {
    STFEM_UNROLL
    for (int y = 0; y < N; ++y)
      STFEM_UNROLL
    for (int x = 0; x < N; ++x) cb_lds[k * N * N + y * N + x] = PA[y * N + x];
  }
  wave_lds_fence();

  // ---- phase B2: collect, integrate z
  STFEM_UNROLL
  for (int y = 0; y < N; ++y)
    STFEM_UNROLL
  for (int z = 0; z < N; ++z) R[y * N + z] += cb_lds[z * N * N + y * N + k];
  plane_sweep<N, +1, true>(prm.eo_SiT, R);
  wave_lds_fence();
  if (out_active) {
    STFEM_UNROLL
    for (int y = 0; y < N; ++y)
      STFEM_UNROLL
    for (int z = 0; z < N; ++z) cb_lds[z * N * N + y * N + k] = R[y * N + z];
  }
  wave_lds_fence();

  // ---- phase A3: integrate y, x
  STFEM_UNROLL
  for (int y = 0; y < N; ++y)
    STFEM_UNROLL
  for (int x = 0; x < N; ++x) PA[y * N + x] = cb_lds[k * N * N + y * N + x];
  plane_sweep<N, +1, false>(prm.eo_SiT, PA);
  plane_sweep<N, +1, true>(prm.eo_SiT, PA);
  wave_lds_fence();
}

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

template <int P, int NBM, bool PLAIN_STORE>
__global__ __launch_bounds__(256) void st_sweep_cart_atomic(const SweepParams prm)
{
  using G = Geometry<P, NBM>;
  constexpr int N = G::N;
  __shared__ double lds_all[G::WAVES * G::LDS_PER_WAVE];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  double *lds = lds_all + wave * G::LDS_PER_WAVE;

  const bool lane_ok = lane < G::ACTIVE;
  const int l = lane_ok ? lane : 0;
  const int k = l % N;
  const int blk = (l / N) % NBM;
  const int cell_in_wave = l / (N * NBM);

  const int64_t wave_global = int64_t(blockIdx.x) * G::WAVES + wave;
  const int64_t cell = wave_global * G::CELLS_PER_WAVE + cell_in_wave;
  const bool cell_ok = lane_ok && cell < prm.ncells;
  const int64_t cc = cell_ok ? cell : 0;
  const int cx = int(cc % prm.ncx), cy = int((cc / prm.ncx) % prm.ncy),
            cz = int(cc / (int64_t(prm.ncx) * prm.ncy));
  const bool in_active = cell_ok && blk < prm.nbi;
  const bool out_active = cell_ok && blk < prm.nbo;

  // temporal coefficients of this lane's output block, with the cell factor folded in
  const double fK = prm.vol * (prm.coef_lap ? prm.coef_lap[cc] : 1.0);
  const double fM = prm.vol * (prm.coef_mass ? prm.coef_mass[cc] : 1.0);
  double aK[NBM], aM[NBM];
  STFEM_UNROLL
  for (int i = 0; i < NBM; ++i) {
    const bool ok = blk < prm.nbo && i < prm.nbi;
    aK[i] = ok ? prm.alpha[blk * prm.nbi + i] * fK : 0.0;
    aM[i] = ok ? prm.beta[blk * prm.nbi + i] * fM : 0.0;
  }

  const PlaneMask pm = plane_mask<P>(prm, cx, cy, cz, k);
  const int64_t base = int64_t(P) * cx + int64_t(prm.nx) * (int64_t(P) * cy + int64_t(prm.ny) * (int64_t(P) * cz + k));

  double PA[N * N];
  {
    const double *s = prm.src[in_active ? blk : 0] + base;
    STFEM_UNROLL
    for (int y = 0; y < N; ++y)
      STFEM_UNROLL
    for (int x = 0; x < N; ++x) {
      const double v = in_active ? s[int64_t(y) * prm.nx + x] : 0.0;
      PA[y * N + x] = constrained<P>(pm, y, x) ? 0.0 : v;
    }
  }

  cell_core<P, NBM>(prm, lds, cell_in_wave, blk, k, in_active, out_active, aK, aM, PA);

  if (out_active) {
    double *d = prm.dst[blk] + base;
    STFEM_UNROLL
    for (int y = 0; y < N; ++y)
      STFEM_UNROLL
    for (int x = 0; x < N; ++x)
      if (!constrained<P>(pm, y, x)) {
        if (PLAIN_STORE) d[int64_t(y) * prm.nx + x] = PA[y * N + x]; // experiment only: wrong on shared DoFs
        else unsafeAtomicAdd(d + int64_t(y) * prm.nx + x, PA[y * N + x]);
      }
  }
}

template <int P, int NBM> int launch_atomic_t(const SweepParams &prm, hipStream_t st)
{
  using G = Geometry<P, NBM>;
  const int64_t cells_per_block = int64_t(G::CELLS_PER_WAVE) * G::WAVES;
  const int64_t blocks = (prm.ncells + cells_per_block - 1) / cells_per_block;
  static const bool plain = getenv("STFEM_EXPERIMENT_PLAIN_STORE") != nullptr;
  if (plain)
    hipLaunchKernelGGL((st_sweep_cart_atomic<P, NBM, true>), dim3((unsigned)blocks), dim3(256), 0, st, prm);
  else
    hipLaunchKernelGGL((st_sweep_cart_atomic<P, NBM, false>), dim3((unsigned)blocks), dim3(256), 0, st, prm);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

constexpr int round_nbm(int nbm) { return nbm <= 4 ? nbm : (nbm <= 6 ? 6 : 8); }

} // namespace

int launch_cart_atomic(int p, const SweepParams &prm, void *stream)
{
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nbm = round_nbm(prm.nbi > prm.nbo ? prm.nbi : prm.nbo);
#define STFEM_CASE(PP, NB) \
  if (p == PP && nbm == NB) return launch_atomic_t<PP, NB>(prm, st);
#define STFEM_CASES(PP) \
  STFEM_CASE(PP, 1) STFEM_CASE(PP, 2) STFEM_CASE(PP, 3) STFEM_CASE(PP, 4) STFEM_CASE(PP, 6) STFEM_CASE(PP, 8)
  STFEM_CASES(1)
  STFEM_CASES(2)
  STFEM_CASES(3)
  STFEM_CASES(4)
#undef STFEM_CASES
#undef STFEM_CASE
  return -2;
}

const char *cart_atomic_name(int, int) { return "st_sweep_cart_atomic"; }

} // namespace stfem
