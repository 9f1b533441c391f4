// "Atomic" variant of the fused space-time cell sweep (gfx950): one wave per group of cells,
// results scattered with global fp64 atomics into a pre-zeroed dst.  Simple and independent of
// the mesh decomposition; kept as the cross-check of the tile variant (stfem_tile.hip) and as
// the STFEM_VARIANT=atomic fallback.  See stfem_core.h for the per-cell algorithm.
#include "stfem_core.h"

#include <hip/hip_runtime.h>

#include <cstdlib>

namespace stfem {
namespace STFEM_PREC {

namespace {

template <int P, int NBM, bool PLAIN_STORE>
__global__ __launch_bounds__(256) void st_sweep_cart_atomic(const SweepParams prm)
{
  using G = Geometry<P, NBM>;
  constexpr int N = G::N;
  __shared__ real_t lds_all[G::WAVES * G::LDS_PER_WAVE];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  real_t *lds = lds_all + wave * G::LDS_PER_WAVE;

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
  const real_t fK = prm.vol * (prm.coef_lap ? prm.coef_lap[cc] : real_t(1));
  const real_t fM = prm.vol * (prm.coef_mass ? prm.coef_mass[cc] : real_t(1));
  real_t aK[NBM], aM[NBM];
  STFEM_UNROLL
  for (int i = 0; i < NBM; ++i) {
    const bool ok = blk < prm.nbo && i < prm.nbi;
    aK[i] = ok ? prm.alpha[blk * prm.nbi + i] * fK : real_t(0);
    aM[i] = ok ? prm.beta[blk * prm.nbi + i] * fM : real_t(0);
  }

  const PlaneMask pm = plane_mask<P>(prm, cx, cy, cz, k);
  const int64_t base = int64_t(P) * cx + int64_t(prm.nx) * (int64_t(P) * cy + int64_t(prm.ny) * (int64_t(P) * cz + k));

  real_t PA[N * N];
  {
    const real_t *s = prm.src[in_active ? blk : 0] + base;
    STFEM_UNROLL
    for (int y = 0; y < N; ++y)
      STFEM_UNROLL
    for (int x = 0; x < N; ++x) {
      const real_t v = in_active ? s[int64_t(y) * prm.nx + x] : real_t(0);
      PA[y * N + x] = constrained<P>(pm, y, x) ? real_t(0) : v;
    }
  }

  cell_core<P, NBM>(prm, lds, cell_in_wave, blk, k, in_active, out_active, aK, aM, PA);

  if (out_active) {
    real_t *d = prm.dst[blk] + base;
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
  (void)hipGetLastError(); // drop a stale sticky error of an unrelated earlier call
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

} // namespace

namespace {

__global__ __launch_bounds__(128) void diagonal_kernel(const DiagParams prm)
{
  const int n = prm.p + 1, n3 = n * n * n;
  const int64_t cell = blockIdx.x;
  const int cx = int(cell % prm.ncx), cy = int((cell / prm.ncx) % prm.ncy), cz = int(cell / (int64_t(prm.ncx) * prm.ncy));
  for (int l = threadIdx.x; l < n3; l += blockDim.x) {
    const int a = l % n, b = (l / n) % n, c = l / (n * n);
    const int ix = prm.p * cx + a, iy = prm.p * cy + b, iz = prm.p * cz + c;
    const bool con = ((prm.dmask & 1) && ix == 0) || ((prm.dmask & 2) && ix == prm.nx - 1) ||
                     ((prm.dmask & 4) && iy == 0) || ((prm.dmask & 8) && iy == prm.ny - 1) ||
                     ((prm.dmask & 16) && iz == 0) || ((prm.dmask & 32) && iz == prm.p * prm.ncz);
    if (con) continue;
    real_t v = real_t(0);
    if (prm.metric) {
      const real_t *m = prm.metric + cell * 8 * n3; // [q][8] records
      for (int qz = 0; qz < n; ++qz)
        for (int qy = 0; qy < n; ++qy)
          for (int qx = 0; qx < n; ++qx) {
            const int q = qx + n * (qy + n * qz);
            const real_t sa = prm.S[qx * n + a], sb = prm.S[qy * n + b], sc = prm.S[qz * n + c];
            const real_t val = sa * sb * sc;
            const real_t g0 = prm.D[qx * n + a] * sb * sc, g1 = sa * prm.D[qy * n + b] * sc, g2 = sa * sb * prm.D[qz * n + c];
            const real_t *mq = m + q * 8;
            v += prm.ms * mq[6] * val * val +
                 prm.ls * (mq[0] * g0 * g0 + mq[3] * g1 * g1 + mq[5] * g2 * g2 +
                           real_t(2) * (mq[1] * g0 * g1 + mq[2] * g0 * g2 + mq[4] * g1 * g2));
          }
    } else {
      const real_t fK = prm.vol * (prm.coef_lap ? prm.coef_lap[cell] : real_t(1));
      const real_t fM = prm.vol * (prm.coef_mass ? prm.coef_mass[cell] : real_t(1));
      v = prm.ms * fM * prm.m1[a] * prm.m1[b] * prm.m1[c] +
          prm.ls * fK * (prm.ihx2 * prm.l1[a] * prm.m1[b] * prm.m1[c] + prm.ihy2 * prm.m1[a] * prm.l1[b] * prm.m1[c] +
                         prm.ihz2 * prm.m1[a] * prm.m1[b] * prm.l1[c]);
    }
    unsafeAtomicAdd(prm.diag + ix + int64_t(prm.nx) * (iy + int64_t(prm.ny) * iz), v);
  }
}

} // namespace

int launch_diagonal(const DiagParams &prm, void *stream)
{
  (void)hipGetLastError(); // drop a stale sticky error of an unrelated earlier call
  const int64_t ncells = int64_t(prm.ncx) * prm.ncy * prm.ncz;
  hipLaunchKernelGGL(diagonal_kernel, dim3((unsigned)ncells), dim3(128), 0, static_cast<hipStream_t>(stream), prm);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

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

} // namespace STFEM_PREC
} // namespace stfem
