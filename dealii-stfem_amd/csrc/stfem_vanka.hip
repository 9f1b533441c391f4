// Cell-patch Vanka / additive-Schwarz smoother of the space-time system.
//
// Replaces PreconditionVanka (reference include/stmg.h:619-907) for the scalar system
// A = Alpha (x) K + Beta (x) M on one rank:
//   set-up (stmg.h:786-829, compute_block_matrix.h:50-139): per cell the block
//       B_c(k + i n, l + j n) = valence(k) * (Beta(i,j) M(k,l) + Alpha(i,j) K(k,l)),   k, l = DoFs of the cell,
//     K, M = the ASSEMBLED matrices with the zero-boundary constraints (tests/tp_01.cc:283-299), inverted by
//     Gauss-Jordan;
//   vmult (stmg.h:832-872): dst = sum over cells of scatter(B_c^-1 gather(src)).
// On the axis-aligned uniform meshes this path serves (cfg 1), the restriction of an assembled Kronecker
// matrix to a cell is the Kronecker product of restricted 1D matrices, and it only depends on which
// neighbours the cell has: at most 27 different blocks per mesh instead of one 250 x 250 block per cell
// (250 kB per Q4 x cG(2) cell in fp32).  The apply is then a GEMM per block class,
//       Y[250 x cells] = B^-1[250 x 250] X[250 x cells],
// with the gather and the scatter fused in: MFMA work (v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32,
// 125 kflop against 4 kB of DoF traffic per cell).  Cells are processed in eight colours (cells of one
// colour share no DoF), so the scatter is plain load-add-store: no atomics, bitwise reproducible.
// General meshes and coefficient tables get one block per cell (the reference's layout): set-up on the host
// from device-computed cell matrices, apply = one HBM-bound block-times-vector per cell (vanka_apply_percell_kernel).
#include "stfem_internal.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <utility>
#include <new>
#include <vector>

#include "stfem_vanka_kernel.h"

namespace {

// ---- general meshes / coefficient tables: one block per cell ----
// Cell matrices from the stored metric: K_c(a,b) = sum_q grad phi_a^T G_q grad phi_b, M_c(a,b) = sum_q Mq phi_a phi_b
// (what MatrixFreeTools::compute_matrix gets from do_cell_integral_local on unit vectors, operators.h:1021-1033).
// One workgroup per cell; a thread takes entries (a, b).  Set-up code: plain, not tuned.
template <typename T>
__global__ __launch_bounds__(256) void vanka_cell_matrices_kernel(int n, const T *__restrict__ metric,
                                                                   const double *__restrict__ S1, const double *__restrict__ D1,
                                                                   double *__restrict__ Kc, double *__restrict__ Mc)
{
  const int nloc = n * n * n;
  extern __shared__ double sm[]; // [nloc][7] metric of the cell, [n*n] S, [n*n] D
  double *met = sm, *S = sm + nloc * 7, *D = S + n * n;
  const long long cell = blockIdx.x;
  for (int i = threadIdx.x; i < nloc * 7; i += 256) met[i] = double(metric[(cell * nloc + i / 7) * 8 + i % 7]);
  for (int i = threadIdx.x; i < n * n; i += 256) { S[i] = S1[i]; D[i] = D1[i]; }
  __syncthreads();
  for (int e = threadIdx.x; e < nloc * nloc; e += 256) {
    const int a = e / nloc, b = e % nloc;
    const int ax = a % n, ay = (a / n) % n, az = a / (n * n), bx = b % n, by = (b / n) % n, bz = b / (n * n);
    double k = 0.0, mm = 0.0;
    for (int qz = 0; qz < n; ++qz)
      for (int qy = 0; qy < n; ++qy)
        for (int qx = 0; qx < n; ++qx) {
          const double *g = met + (qx + n * (qy + n * qz)) * 7;
          const double sax = S[qx * n + ax], say = S[qy * n + ay], saz = S[qz * n + az];
          const double sbx = S[qx * n + bx], sby = S[qy * n + by], sbz = S[qz * n + bz];
          const double ga[3] = {D[qx * n + ax] * say * saz, sax * D[qy * n + ay] * saz, sax * say * D[qz * n + az]};
          const double gb[3] = {D[qx * n + bx] * sby * sbz, sbx * D[qy * n + by] * sbz, sbx * sby * D[qz * n + bz]};
          k += ga[0] * (g[0] * gb[0] + g[1] * gb[1] + g[2] * gb[2]) + ga[1] * (g[1] * gb[0] + g[3] * gb[1] + g[4] * gb[2]) +
               ga[2] * (g[2] * gb[0] + g[4] * gb[1] + g[5] * gb[2]);
          mm += g[6] * sax * say * saz * sbx * sby * sbz;
        }
    Kc[cell * nloc * nloc + e] = k;
    Mc[cell * nloc * nloc + e] = mm;
  }
}

struct VankaCellParams {
  const void *src[VK_MAX_BLOCKS];
  void *dst[VK_MAX_BLOCKS];
  const void *blocks; // [cell][kpad][mpad], element (row r, column k) of the cell's inverse at [k][r]
  const int *off;
  const int *cell;    // [ncell of this colour]: cell number
  int m, mpad, kpad, nloc, p, colour;
  int ncx, ncy, ncz, nx, ny;
  void *flat;         // two-phase apply (small meshes): Y[cell][mpad]; nullptr: colour launches
  double omega;
  int accumulate;
};

// y = B_c^-1 x per cell, the block streamed from HBM once (the reference's apply: stmg.h:845-867): one workgroup per
// cell, thread r = row r; x in LDS.  HBM-bound: kpad * mpad elements per cell against 2 m of DoF traffic.
template <typename T>
__global__ __launch_bounds__(256) void vanka_apply_percell_kernel(const VankaCellParams prm)
{
  __shared__ T xs[VK_MAX_ROWS];
  const int cell = prm.flat ? int(blockIdx.x) : prm.cell[blockIdx.x];
  const int cx = cell % prm.ncx, cy = (cell / prm.ncx) % prm.ncy, cz = cell / (prm.ncx * prm.ncy);
  const long long base = (long long)prm.p * cx + (long long)prm.nx * ((long long)prm.p * cy + (long long)prm.ny * prm.p * cz);
  for (int r = threadIdx.x; r < prm.kpad; r += 256) {
    T v = T(0);
    if (r < prm.m) {
      const int blk = r / prm.nloc, n = r - blk * prm.nloc;
      const T *sp = static_cast<const T *>(prm.src[0]);
#pragma unroll
      for (int b = 1; b < VK_MAX_BLOCKS; ++b)
        if (b == blk) sp = static_cast<const T *>(prm.src[b]);
      v = sp[base + prm.off[n]];
    }
    xs[r] = v;
  }
  __syncthreads();
  const T *B = static_cast<const T *>(prm.blocks) + (size_t)cell * prm.kpad * prm.mpad;
  for (int r = threadIdx.x; r < prm.m; r += 256) {
    T a0 = T(0), a1 = T(0), a2 = T(0), a3 = T(0);
    const T *col = B + r;
    int k = 0;
    for (; k + 3 < prm.kpad; k += 4) {
      a0 = fma(col[(size_t)k * prm.mpad], xs[k], a0);
      a1 = fma(col[(size_t)(k + 1) * prm.mpad], xs[k + 1], a1);
      a2 = fma(col[(size_t)(k + 2) * prm.mpad], xs[k + 2], a2);
      a3 = fma(col[(size_t)(k + 3) * prm.mpad], xs[k + 3], a3);
    }
    for (; k < prm.kpad; ++k) a0 = fma(col[(size_t)k * prm.mpad], xs[k], a0);
    const T y = (a0 + a1) + (a2 + a3);
    if (prm.flat) {
      static_cast<T *>(prm.flat)[size_t(cell) * prm.mpad + r] = y;
      continue;
    }
    const int blk = r / prm.nloc, n = r - blk * prm.nloc;
    T *dp = static_cast<T *>(prm.dst[0]);
#pragma unroll
    for (int b = 1; b < VK_MAX_BLOCKS; ++b)
      if (b == blk) dp = static_cast<T *>(prm.dst[b]);
    // first touch (lowest colour among the cells sharing the DoF) stores, the others add
    const int np = prm.p + 1;
    const int idx[3] = {n % np, (n / np) % np, n / (np * np)};
    const int cc[3] = {cx, cy, cz}, nc[3] = {prm.ncx, prm.ncy, prm.ncz};
    bool first = true;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const bool shared = (idx[d] == 0 && cc[d] > 0) || (idx[d] == prm.p && cc[d] < nc[d] - 1);
      if (shared && ((prm.colour >> d) & 1)) first = false;
    }
    T *q = dp + base + prm.off[n];
    const T oy = T(prm.omega) * y;
    *q = (first && !prm.accumulate) ? oy : *q + oy;
  }
}


// Second phase of the two-phase apply (small meshes, where eight colour launches of ~20 us each are all latency): every DoF sums
// the rows its (up to eight) cells left in the scratch array, in a fixed order (z, y, x of the cells) - reproducible - and
// stores: dst = sum over cells of scatter(B_c^-1 gather(src)) (stmg.h:836-867) in two launches instead of eight.
struct VankaCollectParams {
  void *dst[VK_MAX_BLOCKS];
  const void *y;    // [slot][mpad]
  const int *slot;  // cell -> slot (nullptr: slot = cell)
  int nb, nloc, p, mpad;
  int ncx, ncy, ncz, nx, ny, nz;
  double omega;
  int accumulate;
};
template <typename T> __global__ __launch_bounds__(256) void vanka_collect_kernel(const VankaCollectParams prm)
{
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)prm.nx * prm.ny * prm.nz) return;
  const int ix = int(i % prm.nx), iy = int((i / prm.nx) % prm.ny), iz = int(i / ((long long)prm.nx * prm.ny));
  const int p = prm.p, np = p + 1;
  // per direction: the cells holding node i and its local index there (a vertex node: the cell below with index p, then the cell above with 0)
  int cc[3][2], ll[3][2], cnt[3];
  const int idx[3] = {ix, iy, iz}, nc[3] = {prm.ncx, prm.ncy, prm.ncz};
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int c0 = idx[d] / p, l0 = idx[d] - c0 * p;
    cnt[d] = 0;
    if (l0 == 0) {
      if (c0 > 0) { cc[d][cnt[d]] = c0 - 1; ll[d][cnt[d]] = p; ++cnt[d]; }
      if (c0 < nc[d]) { cc[d][cnt[d]] = c0; ll[d][cnt[d]] = 0; ++cnt[d]; }
    } else {
      cc[d][0] = c0; ll[d][0] = l0; cnt[d] = 1;
    }
  }
  T acc[VK_MAX_BLOCKS];
#pragma unroll
  for (int b = 0; b < VK_MAX_BLOCKS; ++b) acc[b] = T(0);
  const T *Y = static_cast<const T *>(prm.y);
  for (int kz = 0; kz < cnt[2]; ++kz)
    for (int ky = 0; ky < cnt[1]; ++ky)
      for (int kx = 0; kx < cnt[0]; ++kx) {
        const int cell = cc[0][kx] + prm.ncx * (cc[1][ky] + prm.ncy * cc[2][kz]);
        const int n = ll[0][kx] + np * (ll[1][ky] + np * ll[2][kz]);
        const T *row = Y + size_t(prm.slot ? prm.slot[cell] : cell) * prm.mpad + n;
#pragma unroll
        for (int b = 0; b < VK_MAX_BLOCKS; ++b)
          if (b < prm.nb) acc[b] += row[b * prm.nloc];
      }
#pragma unroll
  for (int b = 0; b < VK_MAX_BLOCKS; ++b)
    if (b < prm.nb) {
      T *d = static_cast<T *>(prm.dst[b]) + i;
      *d = prm.accumulate ? *d + T(prm.omega) * acc[b] : T(prm.omega) * acc[b];
    }
}

// ---- device-side set-up of the per-cell blocks (large general meshes: MI355X holds the reference's one-block-per-cell layout of
// a whole configs[2] / configs[3] mesh in HBM; the set-up must then not go through the host either) ----
struct VankaAssembleParams {
  const double *Kc, *Mc; // cell matrices of the cell layers [zw0, zw1)
  double *B;             // [cell of the batch][m][m]
  int ncx, ncy, ncz, p, nb, dmask;
  int zw0;               // first cell layer held in Kc / Mc
  int z0, ncells_batch;  // the batch: cell layers from z0, ncells_batch cells
  double Alpha[VK_MAX_BLOCKS * VK_MAX_BLOCKS], Beta[VK_MAX_BLOCKS * VK_MAX_BLOCKS];
};

// One workgroup per cell: the restriction of the ASSEMBLED K, M to the cell's DoFs (own cell matrix + what the neighbours add on
// shared faces, edges and vertices: compute_block_matrix.h:50-139), zero-boundary rows / columns with the diagonal kept, valence
// scaling of the rows, Kronecker with Alpha / Beta (stmg.h:806-829).  Same steps as vanka_create_per_cell_host.
__global__ __launch_bounds__(256) void vanka_assemble_kernel(const VankaAssembleParams prm)
{
  const int n = prm.p + 1, nloc = n * n * n, m = prm.nb * nloc;
  const int lc = blockIdx.x; // cell of the batch
  const int cpl = prm.ncx * prm.ncy;
  const int cz = prm.z0 + lc / cpl, cy = (lc % cpl) / prm.ncx, cx = lc % prm.ncx;
  const int cc[3] = {cx, cy, cz}, nc[3] = {prm.ncx, prm.ncy, prm.ncz};
  double *B = prm.B + (size_t)lc * m * m;
  for (int e = threadIdx.x; e < nloc * nloc; e += blockDim.x) {
    const int a = e / nloc, b = e % nloc;
    const int ia[3] = {a % n, (a / n) % n, a / (n * n)}, ib[3] = {b % n, (b / n) % n, b / (n * n)};
    double val = 1.0;
    bool con_a = false, con_b = false;
    int lo[3], hi[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if ((ia[d] == 0 && cc[d] > 0) || (ia[d] == prm.p && cc[d] < nc[d] - 1)) val *= 2.0;
      const bool lo_face = cc[d] == 0 && (prm.dmask & (1 << (2 * d))), hi_face = cc[d] == nc[d] - 1 && (prm.dmask & (2 << (2 * d)));
      con_a = con_a || (ia[d] == 0 && lo_face) || (ia[d] == prm.p && hi_face);
      con_b = con_b || (ib[d] == 0 && lo_face) || (ib[d] == prm.p && hi_face);
      lo[d] = (ia[d] == 0 && ib[d] == 0 && cc[d] > 0) ? -1 : 0;
      hi[d] = (ia[d] == prm.p && ib[d] == prm.p && cc[d] < nc[d] - 1) ? 1 : 0;
    }
    double ks = 0.0, ms = 0.0;
    if (a == b || !(con_a || con_b)) {
      for (int sz = lo[2]; sz <= hi[2]; ++sz)
        for (int sy = lo[1]; sy <= hi[1]; ++sy)
          for (int sx = lo[0]; sx <= hi[0]; ++sx) {
            const int sh[3] = {sx, sy, sz};
            int a2 = 0, b2 = 0, mul = 1;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
              a2 += (sh[d] == -1 ? prm.p : (sh[d] == 1 ? 0 : ia[d])) * mul;
              b2 += (sh[d] == -1 ? prm.p : (sh[d] == 1 ? 0 : ib[d])) * mul;
              mul *= n;
            }
            const size_t c2 = (size_t)(cx + sx) + (size_t)prm.ncx * ((cy + sy) + (size_t)prm.ncy * (cz + sz - prm.zw0));
            ks += prm.Kc[(c2 * nloc + a2) * nloc + b2];
            ms += prm.Mc[(c2 * nloc + a2) * nloc + b2];
          }
    }
    for (int i = 0; i < prm.nb; ++i)
      for (int j = 0; j < prm.nb; ++j)
        B[(size_t)(i * nloc + a) * m + j * nloc + b] = val * (prm.Beta[i * prm.nb + j] * ms + prm.Alpha[i * prm.nb + j] * ks);
  }
}

// In-place Gauss-Jordan inverse with partial pivoting (FullMatrix::gauss_jordan, stmg.h:828; the steps of invert() above), one
// workgroup per m x m matrix in global memory, then the block in the apply's layout: out[k][r] = T(inverse(r, k)), row stride mpad.
template <typename T>
__global__ __launch_bounds__(256) void vanka_invert_kernel(double *__restrict__ Ball, T *__restrict__ out_all, int m, int mpad, int kpad,
                                                            long long cell0, int *__restrict__ singular)
{
  __shared__ double rowc[VK_MAX_ROWS];
  __shared__ int piv[VK_MAX_ROWS];
  __shared__ double rbest[256];
  __shared__ int rarg[256];
  double *A = Ball + (size_t)blockIdx.x * m * m;
  const int t = threadIdx.x;
  for (int c = 0; c < m; ++c) {
    double best = -1.0;
    int arg = c;
    for (int r = c + t; r < m; r += 256) {
      const double v = fabs(A[(size_t)r * m + c]);
      if (v > best) { best = v; arg = r; }
    }
    rbest[t] = best;
    rarg[t] = arg;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (t < s) {
        // the first (lowest) row among equal magnitudes, as the sequential search picks it
        if (rbest[t + s] > rbest[t] || (rbest[t + s] == rbest[t] && rarg[t + s] < rarg[t])) { rbest[t] = rbest[t + s]; rarg[t] = rarg[t + s]; }
      }
      __syncthreads();
    }
    const int p = rarg[0];
    const double pv = rbest[0];
    if (pv <= 0.0) {
      if (t == 0) *singular = 1;
      return;
    }
    if (t == 0) piv[c] = p;
    // swap rows c and p, scale the pivot row, keep it in LDS
    const double inv = 1.0 / A[(size_t)p * m + c];
    __syncthreads(); // (every thread has read the pivot before the row is rewritten)
    for (int k = t; k < m; k += 256) {
      const double up = A[(size_t)p * m + k], uc = A[(size_t)c * m + k];
      if (p != c) A[(size_t)p * m + k] = uc;
      const double v = (k == c ? 1.0 : up) * inv;
      A[(size_t)c * m + k] = v;
      rowc[k] = v;
    }
    __syncthreads();
    // eliminate column c from every other row: a wave takes a row at a time, lanes along the row
    const int wave = t >> 6, lane = t & 63;
    for (int r = wave; r < m; r += 4) {
      if (r == c) continue;
      const double f = A[(size_t)r * m + c];
      if (f == 0.0) continue;
      for (int k = lane; k < m; k += 64) {
        const double v = (k == c ? 0.0 : A[(size_t)r * m + k]) - f * rowc[k];
        A[(size_t)r * m + k] = v;
      }
    }
    __syncthreads();
  }
  // undo the row exchanges on the columns, last first
  for (int c = m - 1; c >= 0; --c) {
    const int p = piv[c];
    if (p != c) {
      for (int r = t; r < m; r += 256) {
        const double a = A[(size_t)r * m + c];
        A[(size_t)r * m + c] = A[(size_t)r * m + p];
        A[(size_t)r * m + p] = a;
      }
      __syncthreads();
    }
  }
  __syncthreads();
  T *out = out_all + (size_t)(cell0 + blockIdx.x) * kpad * mpad;
  for (int e = t; e < kpad * mpad; e += 256) {
    const int k = e / mpad, r = e % mpad;
    out[e] = (k < m && r < m) ? T(A[(size_t)r * m + k]) : T(0);
  }
}

// in-place Gauss-Jordan inverse with partial pivoting (FullMatrix::gauss_jordan, stmg.h:828)
bool invert(int n, std::vector<double> &A)
{
  std::vector<int> piv(n);
  for (int c = 0; c < n; ++c) {
    int p = c;
    double best = std::abs(A[size_t(c) * n + c]);
    for (int r = c + 1; r < n; ++r)
      if (std::abs(A[size_t(r) * n + c]) > best) { best = std::abs(A[size_t(r) * n + c]); p = r; }
    if (best == 0.0) return false;
    piv[c] = p;
    if (p != c)
      for (int k = 0; k < n; ++k) std::swap(A[size_t(c) * n + k], A[size_t(p) * n + k]);
    const double inv = 1.0 / A[size_t(c) * n + c];
    A[size_t(c) * n + c] = 1.0;
    for (int k = 0; k < n; ++k) A[size_t(c) * n + k] *= inv;
    for (int r = 0; r < n; ++r) {
      if (r == c) continue;
      const double f = A[size_t(r) * n + c];
      if (f == 0.0) continue;
      A[size_t(r) * n + c] = 0.0;
      for (int k = 0; k < n; ++k) A[size_t(r) * n + k] -= f * A[size_t(c) * n + k];
    }
  }
  for (int c = n - 1; c >= 0; --c)
    if (piv[c] != c)
      for (int r = 0; r < n; ++r) std::swap(A[size_t(r) * n + c], A[size_t(r) * n + piv[c]]);
  return true;
}

thread_local char g_vanka_err[256] = "";

} // namespace

struct stfem_vanka {
  stfem_ctx *ctx = nullptr;
  int nb = 0, nloc = 0, m = 0, mt = 0, mtw = 0, parts = 0, mpad = 0, kpad = 0, nclasses = 0;
  void *d_blocks = nullptr;
  void *d_blocks_base = nullptr; // allocation d_blocks points into (blocks built on an extended context: the ghost layers' come first)
  int *d_off = nullptr;
  int *d_cell[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  int *d_cls[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  int nquad[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool per_cell = false; // one block per cell (general meshes, coefficient tables): d_blocks is [cell][kpad][mpad]
  int ncol[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // cells per colour (d_cell[colour] = their numbers)
  // two-phase apply (meshes of up to VK_FLAT_CELLS cells): all cells in ONE launch, rows to d_flat, then vanka_collect_kernel
  bool flat = false;
  void *d_flat = nullptr;
  int *d_cell_all = nullptr, *d_cls_all = nullptr, *d_slot = nullptr;
  int nquad_all = 0;
};
constexpr long long VK_FLAT_CELLS = 50000; // (36^3 cells: 45 us instead of 8 x 23; at 72^3 the scratch traffic costs more than the launches)
static bool vanka_wants_flat(const stfem_ctx *c)
{
  if (const char *e = getenv("STFEM_VANKA_COLOURS")) // (tests and measurements: the colour launches on small meshes too)
    if (atoi(e)) return false;
  return c->ncells <= VK_FLAT_CELLS;
}

#define VK_TRY(call)                                                                  \
  do {                                                                                \
    hipError_t e_ = (call);                                                           \
    if (e_ != hipSuccess) {                                                           \
      snprintf(g_vanka_err, sizeof(g_vanka_err), "%s: %s", #call, hipGetErrorString(e_)); \
      return STFEM_ERR_HIP;                                                           \
    }                                                                                 \
  } while (0)

template <typename T, int NLOC> static const void *vanka_kernel(int mtw)
{
  switch (mtw) {
#define VK_CASE(MT) case MT: return reinterpret_cast<const void *>(&vanka_apply_kernel<T, NLOC, MT>);
    VK_CASE(1) VK_CASE(2) VK_CASE(3) VK_CASE(4) VK_CASE(6) VK_CASE(8)
#undef VK_CASE
    default: return nullptr;
  }
}
template <typename T> static const void *vanka_kernel(int p, int mtw)
{
  switch (p) {
    case 1: return vanka_kernel<T, 8>(mtw);
    case 2: return vanka_kernel<T, 27>(mtw);
    case 3: return vanka_kernel<T, 64>(mtw);
    case 4: return vanka_kernel<T, 125>(mtw);
    case 5: return vanka_kernel<T, 216>(mtw);
    default: return nullptr;
  }
}
static const void *vanka_kernel(const stfem_ctx *c, int mtw) { return c->prec ? vanka_kernel<float>(c->p, mtw) : vanka_kernel<double>(c->p, mtw); }

static int vanka_launch(const stfem_vanka *v, VankaParams &prm, int nquad, hipStream_t st)
{
  const void *k = vanka_kernel(v->ctx, v->mtw);
  if (!k) return STFEM_ERR_UNSUPPORTED;
  void *args[] = {&prm};
  return hipLaunchKernel(k, dim3(nquad, v->parts), dim3(256), args, 0, st) == hipSuccess ? STFEM_OK : STFEM_ERR_HIP;
}

static int vanka_collect(const stfem_vanka *v, stfem_vec *dst, double omega, int accumulate, hipStream_t st)
{
  const stfem_ctx *c = v->ctx;
  VankaCollectParams cp;
  std::memset(&cp, 0, sizeof(cp));
  for (int i = 0; i < v->nb; ++i) cp.dst[i] = dst->blk[i];
  cp.y = v->d_flat;
  cp.slot = v->d_slot;
  cp.nb = v->nb; cp.nloc = v->nloc; cp.p = c->p; cp.mpad = v->mpad;
  cp.ncx = c->nc[0]; cp.ncy = c->nc[1]; cp.ncz = c->nc[2]; cp.nx = c->nd[0]; cp.ny = c->nd[1]; cp.nz = c->nd[2];
  cp.omega = omega; cp.accumulate = accumulate;
  const unsigned grid = (unsigned)((c->ndofs + 255) / 256);
  if (c->prec) hipLaunchKernelGGL(vanka_collect_kernel<float>, dim3(grid), dim3(256), 0, st, cp);
  else hipLaunchKernelGGL(vanka_collect_kernel<double>, dim3(grid), dim3(256), 0, st, cp);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_vanka_err, sizeof(g_vanka_err), "vanka_collect_kernel: %s", hipGetErrorString(e));
    return STFEM_ERR_HIP;
  }
  return STFEM_OK;
}

// Row tiles (16 rows each) per workgroup: a cell block of `tiles` tiles is split into parts of mtw tiles, one
// workgroup each (smaller parts: more workgroups per launch and per CU; larger: less set-up per MFMA).
// Measured on cfg 1 (16 tiles; profiles/r2/vanka): fp64 1.16 / 1.25 ms with 4 / 8 tiles per workgroup, fp32 0.74 / 0.69;
// two or four 16-cell column batches per wave (one staged slab and one LDS read for 2 - 4 MFMAs) 1.19 - 1.47 ms: slower;
// capping the resident workgroups per CU changes nothing.
static void vanka_plan(stfem_vanka *v, int tiles)
{
  int env_tiles = 0;
  if (const char *e = getenv("STFEM_VANKA_TILES")) env_tiles = atoi(e); // (experiments)
  double best = 1e30;
  const int cand64[] = {4, 8, 6, 3, 2, 1}, cand32[] = {8, 4, 6, 3, 2, 1};
  for (int mtw : (v->ctx->prec ? cand32 : cand64)) {
    if (tiles <= 4 ? mtw != tiles : mtw > tiles) continue; // small blocks: one part
    if (env_tiles && mtw != env_tiles) continue;
    if (!vanka_kernel(v->ctx, mtw)) continue;
    const int parts = (tiles + mtw - 1) / mtw;
    const double cost = double(parts * mtw) / tiles * (mtw >= 4 ? 1.0 : 1.1); // padded row tiles are computed too
    if (cost < best - 1e-9) {
      best = cost;
      v->mtw = mtw; v->parts = parts; v->mt = parts * mtw;
    }
  }
}

// DoF offsets of a cell and the cells of every colour (per-cell variant)
static int vanka_per_cell_tables(stfem_vanka *v)
{
  stfem_ctx *c = v->ctx;
  const int n = c->p + 1, nloc = v->nloc;
  const int ncx = c->nc[0], ncy = c->nc[1], ncz = c->nc[2];
  // ---- DoF offsets and the cells of every colour
  std::vector<int> off(nloc);
  for (int kz = 0; kz < n; ++kz)
    for (int jy = 0; jy < n; ++jy)
      for (int ix = 0; ix < n; ++ix) off[ix + n * (jy + n * kz)] = ix + c->nd[0] * (jy + c->nd[1] * kz);
  if (hipMalloc(&v->d_off, nloc * sizeof(int)) != hipSuccess ||
      hipMemcpy(v->d_off, off.data(), nloc * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
    return STFEM_ERR_HIP;
  for (int colour = 0; colour < 8; ++colour) {
    std::vector<int> cells;
    for (int cz = colour >> 2; cz < ncz; cz += 2)
      for (int cy = (colour >> 1) & 1; cy < ncy; cy += 2)
        for (int cx = colour & 1; cx < ncx; cx += 2) cells.push_back(cx + ncx * (cy + ncy * cz));
    v->ncol[colour] = int(cells.size());
    if (cells.empty()) continue;
    if (hipMalloc(&v->d_cell[colour], cells.size() * sizeof(int)) != hipSuccess ||
        hipMemcpy(v->d_cell[colour], cells.data(), cells.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
      return STFEM_ERR_HIP;
  }
  return STFEM_OK;
}

// two-phase apply of the per-cell variant: the scratch array (slot = cell)
static int vanka_flat_per_cell(stfem_vanka *v)
{
  if (!vanka_wants_flat(v->ctx)) return STFEM_OK;
  if (hipMalloc(&v->d_flat, size_t(v->ctx->ncells) * v->mpad * v->ctx->es) != hipSuccess) return STFEM_ERR_OUT_OF_MEMORY;
  v->flat = true;
  return STFEM_OK;
}

// Set-up of the per-cell blocks (general meshes, coefficient tables): cell matrices on the device from the stored
// metric, then on the host the reference's steps one by one (stmg.h:786-829, compute_block_matrix.h:50-139):
// restriction of the assembled matrices to the cell's DoFs (= the cell's own matrix + what the neighbours add on
// shared faces, edges and vertices), zero-boundary rows / columns, valence scaling, Kronecker with Alpha / Beta,
// Gauss-Jordan.  Meant for the mesh sizes the reference can hold too (one (n_blocks nloc)^2 block per cell).
static int vanka_create_per_cell_host(stfem_vanka *v, const double *Alpha, const double *Beta)
{
  stfem_ctx *c = v->ctx;
  const int p = c->p, n = p + 1, nloc = v->nloc, m = v->m, nb = v->nb;
  v->per_cell = true;
  v->mt = (m + 15) / 16;
  v->mpad = 16 * v->mt;
  v->kpad = ((m + 3) / 4) * 4;
  const size_t bsz = size_t(v->kpad) * v->mpad;
  if (double(c->ncells) * double(bsz) * double(c->es) > 64e9) {
    snprintf(g_vanka_err, sizeof(g_vanka_err), "per-cell blocks of %lld cells need %.1f GB", (long long)c->ncells,
             double(c->ncells) * double(bsz) * double(c->es) * 1e-9);
    return STFEM_ERR_OUT_OF_MEMORY;
  }
  // ---- cell matrices
  const void *metric = nullptr;
  int rc = stfem_internal_metric(c, &metric, nullptr);
  if (rc != STFEM_OK) return rc;
  const size_t nmat = size_t(c->ncells) * nloc * nloc;
  double *d_tab = nullptr, *d_K = nullptr, *d_M = nullptr;
  std::vector<double> tabs(2 * n * n);
  for (int i = 0; i < n * n; ++i) { tabs[i] = c->tab.S[i]; tabs[n * n + i] = c->tab.D[i]; }
  auto cleanup = [&]() {
    if (d_tab) (void)hipFree(d_tab);
    if (d_K) (void)hipFree(d_K);
    if (d_M) (void)hipFree(d_M);
  };
  if (hipMalloc(&d_tab, tabs.size() * sizeof(double)) != hipSuccess || hipMalloc(&d_K, nmat * sizeof(double)) != hipSuccess ||
      hipMalloc(&d_M, nmat * sizeof(double)) != hipSuccess) {
    cleanup();
    return STFEM_ERR_OUT_OF_MEMORY;
  }
  std::vector<double> Kc(nmat), Mc(nmat);
  {
    hipError_t e = hipMemcpy(d_tab, tabs.data(), tabs.size() * sizeof(double), hipMemcpyHostToDevice);
    const size_t lds = (size_t(nloc) * 7 + 2 * n * n) * sizeof(double);
    if (e == hipSuccess) {
      if (c->prec)
        hipLaunchKernelGGL(vanka_cell_matrices_kernel<float>, dim3((unsigned)c->ncells), dim3(256), lds, 0, n,
                           static_cast<const float *>(metric), d_tab, d_tab + n * n, d_K, d_M);
      else
        hipLaunchKernelGGL(vanka_cell_matrices_kernel<double>, dim3((unsigned)c->ncells), dim3(256), lds, 0, n,
                           static_cast<const double *>(metric), d_tab, d_tab + n * n, d_K, d_M);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(Kc.data(), d_K, nmat * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(Mc.data(), d_M, nmat * sizeof(double), hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess) {
      snprintf(g_vanka_err, sizeof(g_vanka_err), "cell matrices: %s", hipGetErrorString(e));
      return STFEM_ERR_HIP;
    }
  }
  // ---- blocks
  const int ncx = c->nc[0], ncy = c->nc[1], ncz = c->nc[2];
  std::vector<double> Kr(size_t(nloc) * nloc), Mr(size_t(nloc) * nloc), B(size_t(m) * m), val(nloc);
  std::vector<char> cn(nloc);
  std::vector<float> out32;
  std::vector<double> out64;
  if (c->prec) out32.assign(size_t(c->ncells) * bsz, 0.0f);
  else out64.assign(size_t(c->ncells) * bsz, 0.0);
  for (int cz = 0; cz < ncz; ++cz)
    for (int cy = 0; cy < ncy; ++cy)
      for (int cx = 0; cx < ncx; ++cx) {
        const int cc[3] = {cx, cy, cz};
        const size_t cell = cx + size_t(ncx) * (cy + size_t(ncy) * cz);
        for (int a = 0; a < nloc; ++a) {
          const int ia[3] = {a % n, (a / n) % n, a / (n * n)};
          double vv = 1.0;
          bool con = false;
          for (int d = 0; d < 3; ++d) {
            if ((ia[d] == 0 && cc[d] > 0) || (ia[d] == p && cc[d] < c->nc[d] - 1)) vv *= 2.0;
            if ((ia[d] == 0 && cc[d] == 0 && (c->dmask & (1 << (2 * d)))) || (ia[d] == p && cc[d] == c->nc[d] - 1 && (c->dmask & (2 << (2 * d)))))
              con = true;
          }
          val[a] = vv;
          cn[a] = con;
          for (int b = 0; b < nloc; ++b) {
            const int ib[3] = {b % n, (b / n) % n, b / (n * n)};
            // cells holding both DoFs: per direction the cell itself, and the neighbour across a face both lie on
            int lo[3], hi[3];
            for (int d = 0; d < 3; ++d) {
              lo[d] = (ia[d] == 0 && ib[d] == 0 && cc[d] > 0) ? -1 : 0;
              hi[d] = (ia[d] == p && ib[d] == p && cc[d] < c->nc[d] - 1) ? 1 : 0;
            }
            double ks = 0.0, ms = 0.0;
            for (int sz = lo[2]; sz <= hi[2]; ++sz)
              for (int sy = lo[1]; sy <= hi[1]; ++sy)
                for (int sx = lo[0]; sx <= hi[0]; ++sx) {
                  const int sh[3] = {sx, sy, sz};
                  int a2 = 0, b2 = 0, mul = 1;
                  for (int d = 0; d < 3; ++d) {
                    const int ja = sh[d] == -1 ? p : (sh[d] == 1 ? 0 : ia[d]), jb = sh[d] == -1 ? p : (sh[d] == 1 ? 0 : ib[d]);
                    a2 += ja * mul;
                    b2 += jb * mul;
                    mul *= n;
                  }
                  const size_t c2 = (cx + sx) + size_t(ncx) * ((cy + sy) + size_t(ncy) * (cz + sz));
                  ks += Kc[(c2 * nloc + a2) * nloc + b2];
                  ms += Mc[(c2 * nloc + a2) * nloc + b2];
                }
            Kr[size_t(a) * nloc + b] = ks;
            Mr[size_t(a) * nloc + b] = ms;
          }
        }
        for (int r = 0; r < nloc; ++r)
          if (cn[r])
            for (int q = 0; q < nloc; ++q)
              if (q != r) {
                Kr[size_t(r) * nloc + q] = Kr[size_t(q) * nloc + r] = 0.0;
                Mr[size_t(r) * nloc + q] = Mr[size_t(q) * nloc + r] = 0.0;
              }
        for (int i = 0; i < nb; ++i)
          for (int j = 0; j < nb; ++j)
            for (int r = 0; r < nloc; ++r)
              for (int q = 0; q < nloc; ++q)
                B[size_t(i * nloc + r) * m + j * nloc + q] =
                  val[r] * (Beta[i * nb + j] * Mr[size_t(r) * nloc + q] + Alpha[i * nb + j] * Kr[size_t(r) * nloc + q]);
        if (!invert(m, B)) {
          snprintf(g_vanka_err, sizeof(g_vanka_err), "singular cell block (cell %zu)", cell);
          return STFEM_ERR_INVALID_ARGUMENT;
        }
        for (int r = 0; r < m; ++r)
          for (int k = 0; k < m; ++k) {
            if (c->prec) out32[cell * bsz + size_t(k) * v->mpad + r] = float(B[size_t(r) * m + k]);
            else out64[cell * bsz + size_t(k) * v->mpad + r] = B[size_t(r) * m + k];
          }
      }
  const void *hostp = c->prec ? static_cast<const void *>(out32.data()) : static_cast<const void *>(out64.data());
  if (hipMalloc(&v->d_blocks, size_t(c->ncells) * bsz * c->es) != hipSuccess) return STFEM_ERR_OUT_OF_MEMORY;
  if (hipMemcpy(v->d_blocks, hostp, size_t(c->ncells) * bsz * c->es, hipMemcpyHostToDevice) != hipSuccess) return STFEM_ERR_HIP;
  v->nclasses = int(c->ncells);
  return vanka_per_cell_tables(v);
}

// The per-cell blocks built on the device, a few cell layers at a time: cell matrices of the layers and their neighbours
// (vanka_cell_matrices_kernel), assembly of the restricted, weighted, Kronecker-combined block (vanka_assemble_kernel), batched
// Gauss-Jordan (vanka_invert_kernel) straight into the apply's layout.  Nothing but Alpha / Beta crosses the host.
static int vanka_create_per_cell_device(stfem_vanka *v, const double *Alpha, const double *Beta)
{
  stfem_ctx *c = v->ctx;
  const int p = c->p, n = p + 1, nloc = v->nloc, m = v->m, nb = v->nb;
  v->per_cell = true;
  v->mt = (m + 15) / 16;
  v->mpad = 16 * v->mt;
  v->kpad = ((m + 3) / 4) * 4;
  const size_t bsz = size_t(v->kpad) * v->mpad;
  const int ncx = c->nc[0], ncy = c->nc[1], ncz = c->nc[2];
  const size_t cpl = size_t(ncx) * ncy;
  const size_t km_layer = 2 * cpl * nloc * nloc * sizeof(double), b_layer = cpl * size_t(m) * m * sizeof(double);
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
  const double need = double(c->ncells) * double(bsz) * double(c->es);
  // scratch of one batch of L cell layers: (L + 2) layers of cell matrices + L layers of fp64 blocks; at most 6 GB,
  // and never more than what is left beside the blocks themselves
  const double budget = std::min(6e9, 0.9 * double(free_b) - need);
  if (budget < 3.0 * double(km_layer) + double(b_layer)) {
    snprintf(g_vanka_err, sizeof(g_vanka_err), "per-cell blocks of %lld cells need %.1f GB (%.1f GB free)", (long long)c->ncells, need * 1e-9, double(free_b) * 1e-9);
    return STFEM_ERR_OUT_OF_MEMORY;
  }
  int L = int((budget - 2.0 * double(km_layer)) / double(km_layer + b_layer));
  L = std::max(1, std::min(L, ncz));
  const void *metric = nullptr;
  int rc = stfem_internal_metric(c, &metric, nullptr);
  if (rc != STFEM_OK) return rc;
  double *d_tab = nullptr, *d_K = nullptr, *d_M = nullptr, *d_B = nullptr;
  int *d_flag = nullptr;
  auto cleanup = [&]() {
    (void)hipFree(d_tab);
    (void)hipFree(d_K);
    (void)hipFree(d_M);
    (void)hipFree(d_B);
    (void)hipFree(d_flag);
  };
  std::vector<double> tabs(2 * n * n);
  for (int i = 0; i < n * n; ++i) { tabs[i] = c->tab.S[i]; tabs[n * n + i] = c->tab.D[i]; }
  const size_t win_cells = cpl * size_t(std::min(ncz, L + 2));
  if (hipMalloc(&v->d_blocks, size_t(c->ncells) * bsz * c->es) != hipSuccess) return STFEM_ERR_OUT_OF_MEMORY;
  if (hipMalloc(&d_tab, tabs.size() * sizeof(double)) != hipSuccess || hipMalloc(&d_K, win_cells * nloc * nloc * sizeof(double)) != hipSuccess ||
      hipMalloc(&d_M, win_cells * nloc * nloc * sizeof(double)) != hipSuccess || hipMalloc(&d_B, cpl * L * size_t(m) * m * sizeof(double)) != hipSuccess ||
      hipMalloc(&d_flag, sizeof(int)) != hipSuccess) {
    cleanup();
    return STFEM_ERR_OUT_OF_MEMORY;
  }
  hipError_t e = hipMemcpy(d_tab, tabs.data(), tabs.size() * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemset(d_flag, 0, sizeof(int));
  VankaAssembleParams ap;
  ap.Kc = d_K; ap.Mc = d_M; ap.B = d_B;
  ap.ncx = ncx; ap.ncy = ncy; ap.ncz = ncz; ap.p = p; ap.nb = nb; ap.dmask = c->dmask;
  for (int i = 0; i < nb * nb; ++i) { ap.Alpha[i] = Alpha[i]; ap.Beta[i] = Beta[i]; }
  const size_t lds = (size_t(nloc) * 7 + 2 * n * n) * sizeof(double);
  for (int z0 = 0; z0 < ncz && e == hipSuccess; z0 += L) {
    const int z1 = std::min(ncz, z0 + L), zw0 = std::max(0, z0 - 1), zw1 = std::min(ncz, z1 + 1);
    const size_t wcells = cpl * size_t(zw1 - zw0), bcells = cpl * size_t(z1 - z0);
    const size_t moff = cpl * size_t(zw0) * nloc * 8; // metric records [cell][q][8]
    if (c->prec)
      hipLaunchKernelGGL(vanka_cell_matrices_kernel<float>, dim3((unsigned)wcells), dim3(256), lds, 0, n, static_cast<const float *>(metric) + moff, d_tab,
                         d_tab + n * n, d_K, d_M);
    else
      hipLaunchKernelGGL(vanka_cell_matrices_kernel<double>, dim3((unsigned)wcells), dim3(256), lds, 0, n, static_cast<const double *>(metric) + moff, d_tab,
                         d_tab + n * n, d_K, d_M);
    ap.zw0 = zw0; ap.z0 = z0; ap.ncells_batch = int(bcells);
    hipLaunchKernelGGL(vanka_assemble_kernel, dim3((unsigned)bcells), dim3(256), 0, 0, ap);
    const long long cell0 = (long long)cpl * z0;
    if (c->prec)
      hipLaunchKernelGGL(vanka_invert_kernel<float>, dim3((unsigned)bcells), dim3(256), 0, 0, d_B, static_cast<float *>(v->d_blocks), m, v->mpad, v->kpad, cell0, d_flag);
    else
      hipLaunchKernelGGL(vanka_invert_kernel<double>, dim3((unsigned)bcells), dim3(256), 0, 0, d_B, static_cast<double *>(v->d_blocks), m, v->mpad, v->kpad, cell0,
                         d_flag);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize(); // (the next batch reuses the scratch)
  }
  int flag = 0;
  if (e == hipSuccess) e = hipMemcpy(&flag, d_flag, sizeof(int), hipMemcpyDeviceToHost);
  cleanup();
  if (e != hipSuccess) {
    snprintf(g_vanka_err, sizeof(g_vanka_err), "per-cell block set-up: %s", hipGetErrorString(e));
    return STFEM_ERR_HIP;
  }
  if (flag) {
    snprintf(g_vanka_err, sizeof(g_vanka_err), "singular cell block");
    return STFEM_ERR_INVALID_ARGUMENT;
  }
  v->nclasses = int(c->ncells);
  return vanka_per_cell_tables(v);
}

static int vanka_create_per_cell(stfem_vanka *v, const double *Alpha, const double *Beta)
{
  const char *e = getenv("STFEM_VANKA_HOST_SETUP"); // the same steps on the host, block by block (for comparison; small meshes only)
  return (e && atoi(e) != 0) ? vanka_create_per_cell_host(v, Alpha, Beta) : vanka_create_per_cell_device(v, Alpha, Beta);
}

extern "C" {

const char *stfem_vanka_last_error(void) { return g_vanka_err; }

int stfem_vanka_create(stfem_ctx *c, int nb, const double *Alpha, const double *Beta, stfem_vanka **out)
{
  return stfem_vanka_create_partitioned(c, nb, Alpha, Beta, 0, out);
}

int stfem_vanka_create_partitioned(stfem_ctx *c, int nb, const double *Alpha, const double *Beta, int neighbour_mask, stfem_vanka **out)
{
  if (!c || !Alpha || !Beta || !out || nb < 1 || nb > VK_MAX_BLOCKS || (neighbour_mask & ~63) || (neighbour_mask & c->dmask))
    return STFEM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  const int p = c->p, n = p + 1, nloc = n * n * n, m = nb * nloc;
  if (m > VK_MAX_ROWS) return STFEM_ERR_UNSUPPORTED; // Q4 with more than 4 temporal blocks, Q5 with more than 2
  VK_TRY(hipSetDevice(c->device));
  stfem_vanka *v = new (std::nothrow) stfem_vanka;
  if (!v) return STFEM_ERR_OUT_OF_MEMORY;
  v->ctx = c; v->nb = nb; v->nloc = nloc; v->m = m;
  // one block per neighbour pattern needs identical cells: axis-aligned uniform mesh, no coefficient tables;
  // everything else gets one block per cell
  if (!c->cartesian || c->coef_layout[0] != 0 || c->coef_layout[1] != 0) {
    if (neighbour_mask) { // (the blocks of the interface cells need the cell matrices of the neighbour rank's cells)
      delete v;
      return STFEM_ERR_UNSUPPORTED;
    }
    int rc = vanka_create_per_cell(v, Alpha, Beta);
    if (rc == STFEM_OK) rc = vanka_flat_per_cell(v);
    if (rc != STFEM_OK) {
      stfem_vanka_destroy(v);
      return rc;
    }
    *out = v;
    return STFEM_OK;
  }

  // 1D nodal matrices of the reference cell: Mhat = S^T W S, Khat = D^T W D
  const stfem::ShapeTables &tab = c->tab;
  std::vector<double> Mh(n * n, 0.0), Kh(n * n, 0.0);
  for (int a = 0; a < n; ++a)
    for (int b = 0; b < n; ++b)
      for (int q = 0; q < n; ++q) {
        Mh[a * n + b] += tab.wq[q] * tab.S[q * n + a] * tab.S[q * n + b];
        Kh[a * n + b] += tab.wq[q] * tab.D[q * n + a] * tab.D[q * n + b];
      }
  // classes: per direction bit 0 = has a lower neighbour, bit 1 = has an upper neighbour - on this rank or, across a face of
  // neighbour_mask, on the rank next to it (valence and assembled entries count those cells too; what they add to the shared
  // DoFs arrives with the caller's add-exchange of the interface planes).  local_class: neighbours on this rank only - the
  // first-touch rule of the scatter.
  auto local_class = [&](int d, int cd) { return (cd > 0 ? 1 : 0) | (cd < c->nc[d] - 1 ? 2 : 0); };
  auto dir_class = [&](int d, int cd) {
    return local_class(d, cd) | ((cd == 0 && (neighbour_mask & (1 << (2 * d)))) ? 1 : 0) | ((cd == c->nc[d] - 1 && (neighbour_mask & (2 << (2 * d)))) ? 2 : 0);
  };
  std::map<int, int> class_id;
  std::vector<int> class_key;
  for (int cz = 0; cz < c->nc[2]; ++cz)
    for (int cy = 0; cy < c->nc[1]; ++cy)
      for (int cx = 0; cx < c->nc[0]; ++cx) {
        const int key = dir_class(0, cx) | (dir_class(1, cy) << 2) | (dir_class(2, cz) << 4);
        if (!class_id.count(key)) {
          class_id[key] = int(class_key.size());
          class_key.push_back(key);
        }
      }
  v->nclasses = int(class_key.size());
  {
    vanka_plan(v, (m + 15) / 16);
    if (v->mtw == 0) {
      delete v;
      return STFEM_ERR_UNSUPPORTED;
    }
    v->mpad = 16 * v->mt;
    v->kpad = ((m + KS - 1) / KS) * KS;
  }
  const size_t bsz = size_t(v->kpad) * v->mpad;
  std::vector<double> all(bsz * v->nclasses, 0.0);
  for (int ci = 0; ci < v->nclasses; ++ci) {
    const int key = class_key[ci];
    // restricted assembled 1D matrices: the end nodes also carry the neighbour's diagonal entry
    std::vector<double> M1[3], K1[3];
    bool con[3][8], shared[3][8];
    for (int d = 0; d < 3; ++d) {
      const int k = (key >> (2 * d)) & 3;
      const double h = c->h[d];
      M1[d].assign(n * n, 0.0);
      K1[d].assign(n * n, 0.0);
      for (int e = 0; e < n * n; ++e) {
        M1[d][e] = h * Mh[e];
        K1[d][e] = Kh[e] / h;
      }
      if (k & 1) { M1[d][0] += h * Mh[p * n + p]; K1[d][0] += Kh[p * n + p] / h; }
      if (k & 2) { M1[d][p * n + p] += h * Mh[0]; K1[d][p * n + p] += Kh[0] / h; }
      for (int a = 0; a < n; ++a) {
        shared[d][a] = (a == 0 && (k & 1)) || (a == p && (k & 2));
        con[d][a] = (a == 0 && !(k & 1) && (c->dmask & (1 << (2 * d)))) || (a == p && !(k & 2) && (c->dmask & (2 << (2 * d))));
      }
    }
    std::vector<double> Kr(size_t(nloc) * nloc), Mr(size_t(nloc) * nloc), val(nloc);
    std::vector<char> cn(nloc);
    for (int kz = 0; kz < n; ++kz)
      for (int jy = 0; jy < n; ++jy)
        for (int ix = 0; ix < n; ++ix) {
          const int r = ix + n * (jy + n * kz);
          val[r] = (shared[0][ix] ? 2.0 : 1.0) * (shared[1][jy] ? 2.0 : 1.0) * (shared[2][kz] ? 2.0 : 1.0);
          cn[r] = con[0][ix] || con[1][jy] || con[2][kz];
          for (int kz2 = 0; kz2 < n; ++kz2)
            for (int jy2 = 0; jy2 < n; ++jy2)
              for (int ix2 = 0; ix2 < n; ++ix2) {
                const int s = ix2 + n * (jy2 + n * kz2);
                const double mx = M1[0][ix * n + ix2], my = M1[1][jy * n + jy2], mz = M1[2][kz * n + kz2];
                const double kx = K1[0][ix * n + ix2], ky = K1[1][jy * n + jy2], kz1 = K1[2][kz * n + kz2];
                Mr[size_t(r) * nloc + s] = mz * my * mx;
                Kr[size_t(r) * nloc + s] = mz * my * kx + mz * ky * mx + kz1 * my * mx;
              }
        }
    // zero-boundary constraints: row and column dropped, the diagonal of the unconstrained assembly stays
    for (int r = 0; r < nloc; ++r)
      if (cn[r])
        for (int s = 0; s < nloc; ++s)
          if (s != r) {
            Kr[size_t(r) * nloc + s] = Kr[size_t(s) * nloc + r] = 0.0;
            Mr[size_t(r) * nloc + s] = Mr[size_t(s) * nloc + r] = 0.0;
          }
    std::vector<double> B(size_t(m) * m);
    for (int i = 0; i < nb; ++i)
      for (int j = 0; j < nb; ++j)
        for (int r = 0; r < nloc; ++r)
          for (int s = 0; s < nloc; ++s)
            B[size_t(i * nloc + r) * m + j * nloc + s] =
              val[r] * (Beta[i * nb + j] * Mr[size_t(r) * nloc + s] + Alpha[i * nb + j] * Kr[size_t(r) * nloc + s]);
    if (!invert(m, B)) {
      snprintf(g_vanka_err, sizeof(g_vanka_err), "singular cell block (class %d)", key);
      delete v;
      return STFEM_ERR_INVALID_ARGUMENT;
    }
    double *dstb = all.data() + bsz * ci;
    for (int r = 0; r < m; ++r)
      for (int k = 0; k < m; ++k) dstb[size_t(k) * v->mpad + r] = B[size_t(r) * m + k];
  }
  // upload in the context's Number type
  if (c->prec) {
    std::vector<float> f(all.size());
    for (size_t i = 0; i < all.size(); ++i) f[i] = float(all[i]);
    if (hipMalloc(&v->d_blocks, f.size() * sizeof(float)) != hipSuccess) { delete v; return STFEM_ERR_OUT_OF_MEMORY; }
    if (hipMemcpy(v->d_blocks, f.data(), f.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { stfem_vanka_destroy(v); return STFEM_ERR_HIP; }
  } else {
    if (hipMalloc(&v->d_blocks, all.size() * sizeof(double)) != hipSuccess) { delete v; return STFEM_ERR_OUT_OF_MEMORY; }
    if (hipMemcpy(v->d_blocks, all.data(), all.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { stfem_vanka_destroy(v); return STFEM_ERR_HIP; }
  }
  std::vector<int> off(nloc);
  for (int kz = 0; kz < n; ++kz)
    for (int jy = 0; jy < n; ++jy)
      for (int ix = 0; ix < n; ++ix) off[ix + n * (jy + n * kz)] = ix + c->nd[0] * (jy + c->nd[1] * kz);
  if (hipMalloc(&v->d_off, nloc * sizeof(int)) != hipSuccess ||
      hipMemcpy(v->d_off, off.data(), nloc * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
    stfem_vanka_destroy(v);
    return STFEM_ERR_HIP;
  }
  // cell lists: per colour, grouped by class into batches of 16 cells, four batches of one class per workgroup
  for (int colour = 0; colour < 8; ++colour) {
    std::map<std::pair<int, int>, std::vector<int>> by_class; // (block class, local neighbour pattern) -> cells
    for (int cz = colour >> 2; cz < c->nc[2]; cz += 2)
      for (int cy = (colour >> 1) & 1; cy < c->nc[1]; cy += 2)
        for (int cx = colour & 1; cx < c->nc[0]; cx += 2) {
          const int key = dir_class(0, cx) | (dir_class(1, cy) << 2) | (dir_class(2, cz) << 4);
          const int local = local_class(0, cx) | (local_class(1, cy) << 2) | (local_class(2, cz) << 4);
          by_class[{class_id[key], local}].push_back(p * cx + c->nd[0] * (p * cy + c->nd[1] * p * cz));
        }
    std::vector<int> cells, cls;
    for (auto &kv : by_class) {
      std::vector<int> &l = kv.second;
      l.resize(((l.size() + 63) / 64) * 64, -1);
      for (size_t q = 0; q < l.size() / 64; ++q) cls.push_back(kv.first.first | (kv.first.second << 8));
      cells.insert(cells.end(), l.begin(), l.end());
    }
    v->nquad[colour] = int(cls.size());
    if (cls.empty()) continue;
    if (hipMalloc(&v->d_cell[colour], cells.size() * sizeof(int)) != hipSuccess ||
        hipMalloc(&v->d_cls[colour], cls.size() * sizeof(int)) != hipSuccess ||
        hipMemcpy(v->d_cell[colour], cells.data(), cells.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(v->d_cls[colour], cls.data(), cls.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
      stfem_vanka_destroy(v);
      return STFEM_ERR_HIP;
    }
  }
  if (vanka_wants_flat(c)) { // two-phase apply: all cells in one launch, grouped by class; cell -> slot for the second phase
    std::map<int, std::vector<std::pair<int, int>>> by_class; // class -> (first DoF, cell number)
    for (int cz = 0; cz < c->nc[2]; ++cz)
      for (int cy = 0; cy < c->nc[1]; ++cy)
        for (int cx = 0; cx < c->nc[0]; ++cx) {
          const int key = dir_class(0, cx) | (dir_class(1, cy) << 2) | (dir_class(2, cz) << 4);
          by_class[class_id[key]].push_back({p * cx + c->nd[0] * (p * cy + c->nd[1] * p * cz), cx + c->nc[0] * (cy + c->nc[1] * cz)});
        }
    std::vector<int> cells, cls, slot(size_t(c->ncells), 0);
    for (auto &kv : by_class) {
      for (const auto &e : kv.second) {
        slot[e.second] = int(cells.size());
        cells.push_back(e.first);
      }
      cells.resize(((cells.size() + 63) / 64) * 64, -1);
      while (cls.size() < cells.size() / 64) cls.push_back(kv.first);
    }
    v->nquad_all = int(cls.size());
    if (hipMalloc(&v->d_cell_all, cells.size() * sizeof(int)) != hipSuccess || hipMalloc(&v->d_cls_all, cls.size() * sizeof(int)) != hipSuccess ||
        hipMalloc(&v->d_slot, slot.size() * sizeof(int)) != hipSuccess || hipMalloc(&v->d_flat, cells.size() * v->mpad * c->es) != hipSuccess ||
        hipMemcpy(v->d_cell_all, cells.data(), cells.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(v->d_cls_all, cls.data(), cls.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(v->d_slot, slot.data(), slot.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
      stfem_vanka_destroy(v);
      return STFEM_ERR_HIP;
    }
    v->flat = true;
  }
  *out = v;
  return STFEM_OK;
}

// One block per cell on a z-slab of a partitioned GENERAL mesh (perturbed cells, coefficient tables: BASELINE configs[2] on more
// than one rank).  The block of a cell next to the interface needs the cell matrices of the neighbour rank's cells
// (stmg.h:688-689, 795-796 and compute_block_matrix.h:68-73: the reference builds the blocks on locally owned AND ghost cells):
// the caller hands over a second context, `extended`, of the same slab plus one ghost cell layer on every side that has a
// neighbour (the vertices of those layers are all that crosses the ranks, once).  The blocks are built on it as on any mesh and
// the ones of the slab's own cells are kept; the apply runs on the slab and leaves PARTIAL sums in the interface planes like
// the uniform-mesh variant (stfem_vanka_create_partitioned).
int stfem_vanka_create_partitioned_general(stfem_ctx *slab, stfem_ctx *extended, int nb, const double *Alpha, const double *Beta,
                                           int neighbour_mask, stfem_vanka **out)
{
  if (!slab || !extended || !Alpha || !Beta || !out || nb < 1) return STFEM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (neighbour_mask & ~(16 | 32)) return STFEM_ERR_UNSUPPORTED; // z-slabs
  const int glo = (neighbour_mask & 16) ? 1 : 0, ghi = (neighbour_mask & 32) ? 1 : 0;
  if (extended->p != slab->p || extended->prec != slab->prec || extended->device != slab->device || extended->nc[0] != slab->nc[0] ||
      extended->nc[1] != slab->nc[1] || extended->nc[2] != slab->nc[2] + glo + ghi)
    return STFEM_ERR_SHAPE_MISMATCH;
  // the extended context carries the domain's constraints: none on a z face with a ghost layer, the slab's elsewhere
  if ((extended->dmask & 15) != (slab->dmask & 15) || (glo && (extended->dmask & 16)) || (ghi && (extended->dmask & 32)) ||
      (!glo && (extended->dmask & 16) != (slab->dmask & 16)) || (!ghi && (extended->dmask & 32) != (slab->dmask & 32)))
    return STFEM_ERR_INVALID_ARGUMENT;
  stfem_vanka *ve = nullptr;
  int rc = stfem_vanka_create_partitioned(extended, nb, Alpha, Beta, 0, &ve);
  if (rc != STFEM_OK) return rc;
  if (!ve->per_cell) { // an axis-aligned uniform mesh: the class variant handles the partition by itself
    stfem_vanka_destroy(ve);
    return stfem_vanka_create_partitioned(slab, nb, Alpha, Beta, neighbour_mask, out);
  }
  stfem_vanka *v = new (std::nothrow) stfem_vanka;
  if (!v) {
    stfem_vanka_destroy(ve);
    return STFEM_ERR_OUT_OF_MEMORY;
  }
  v->ctx = slab; v->nb = ve->nb; v->nloc = ve->nloc; v->m = ve->m;
  v->mt = ve->mt; v->mtw = ve->mtw; v->parts = ve->parts; v->mpad = ve->mpad; v->kpad = ve->kpad;
  v->per_cell = true;
  v->nclasses = int(slab->ncells);
  // take the allocation over; the slab's cells start behind the lower ghost layer
  v->d_blocks_base = ve->d_blocks;
  ve->d_blocks = nullptr;
  const size_t bsz = size_t(v->kpad) * v->mpad;
  v->d_blocks = static_cast<char *>(v->d_blocks_base) + size_t(glo) * size_t(slab->nc[0]) * slab->nc[1] * bsz * slab->es;
  stfem_vanka_destroy(ve);
  VK_TRY(hipSetDevice(slab->device));
  rc = vanka_per_cell_tables(v);
  if (rc == STFEM_OK) rc = vanka_flat_per_cell(v);
  if (rc != STFEM_OK) {
    stfem_vanka_destroy(v);
    return rc;
  }
  *out = v;
  return STFEM_OK;
}

void stfem_vanka_destroy(stfem_vanka *v)
{
  if (!v) return;
  (void)hipSetDevice(v->ctx->device);
  if (v->d_blocks_base) (void)hipFree(v->d_blocks_base);
  else if (v->d_blocks) (void)hipFree(v->d_blocks);
  if (v->d_off) (void)hipFree(v->d_off);
  if (v->d_flat) (void)hipFree(v->d_flat);
  if (v->d_cell_all) (void)hipFree(v->d_cell_all);
  if (v->d_cls_all) (void)hipFree(v->d_cls_all);
  if (v->d_slot) (void)hipFree(v->d_slot);
  for (int i = 0; i < 8; ++i) {
    if (v->d_cell[i]) (void)hipFree(v->d_cell[i]);
    if (v->d_cls[i]) (void)hipFree(v->d_cls[i]);
  }
  delete v;
}

int stfem_vanka_n_classes(const stfem_vanka *v) { return v ? v->nclasses : 0; }
int stfem_vanka_plan(const stfem_vanka *v, int32_t out[2])
{
  if (!v || !out) return STFEM_ERR_INVALID_ARGUMENT;
  out[0] = v->mtw; out[1] = v->parts;
  return STFEM_OK;
}

int stfem_vanka_vmult(stfem_vanka *v, stfem_vec *dst, const stfem_vec *src, void *stream) { return stfem_vanka_step(v, dst, 1.0, 0, src, stream); }

// dst = (accumulate ? dst : 0) + omega * V src: the relaxation step x <- x + omega P^-1 r of the multigrid smoothers
// (PreconditionRelaxation around the Vanka smoother, stmg.h:1199-1238) without a temporary vector and a separate update pass
int stfem_vanka_step(stfem_vanka *v, stfem_vec *dst, double omega, int accumulate, const stfem_vec *src, void *stream)
{
  if (!v || !dst || !src) return STFEM_ERR_INVALID_ARGUMENT;
  struct Scope { // the reference's TimerOutput scope "vanka" (stmg.h:835)
    Scope() { stfem_trace_push("vanka"); }
    ~Scope() { stfem_trace_pop(); }
  } scope;
  if (dst->ctx != v->ctx || src->ctx != v->ctx || dst->nb != v->nb || src->nb != v->nb) return STFEM_ERR_SHAPE_MISMATCH;
  for (int i = 0; i < v->nb; ++i)
    for (int j = 0; j < v->nb; ++j)
      if (dst->blk[i] == src->blk[j]) return STFEM_ERR_ALIAS;
  stfem_ctx *c = v->ctx;
  VK_TRY(hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  // dst = 0 (stmg.h:836) is not a pass of its own: the first cell to touch a DoF stores (every DoF has one)
  if (v->per_cell) {
    VankaCellParams cp;
    std::memset(&cp, 0, sizeof(cp));
    for (int i = 0; i < v->nb; ++i) {
      cp.src[i] = src->blk[i];
      cp.dst[i] = dst->blk[i];
    }
    cp.blocks = v->d_blocks;
    cp.off = v->d_off;
    cp.m = v->m; cp.mpad = v->mpad; cp.kpad = v->kpad; cp.nloc = v->nloc; cp.p = c->p;
    cp.ncx = c->nc[0]; cp.ncy = c->nc[1]; cp.ncz = c->nc[2]; cp.nx = c->nd[0]; cp.ny = c->nd[1];
    cp.omega = omega; cp.accumulate = accumulate;
    (void)hipGetLastError();
    if (v->flat) {
      cp.flat = v->d_flat;
      if (c->prec) hipLaunchKernelGGL(vanka_apply_percell_kernel<float>, dim3((unsigned)c->ncells), dim3(256), 0, st, cp);
      else hipLaunchKernelGGL(vanka_apply_percell_kernel<double>, dim3((unsigned)c->ncells), dim3(256), 0, st, cp);
      return vanka_collect(v, dst, omega, accumulate, st);
    }
    for (int colour = 0; colour < 8; ++colour) {
      if (v->ncol[colour] == 0) continue;
      cp.cell = v->d_cell[colour];
      cp.colour = colour;
      if (c->prec) hipLaunchKernelGGL(vanka_apply_percell_kernel<float>, dim3(v->ncol[colour]), dim3(256), 0, st, cp);
      else hipLaunchKernelGGL(vanka_apply_percell_kernel<double>, dim3(v->ncol[colour]), dim3(256), 0, st, cp);
      const hipError_t e = hipGetLastError();
      if (e != hipSuccess) {
        snprintf(g_vanka_err, sizeof(g_vanka_err), "vanka_apply_percell_kernel: %s", hipGetErrorString(e));
        return STFEM_ERR_HIP;
      }
    }
    return STFEM_OK;
  }
  VankaParams prm;
  std::memset(&prm, 0, sizeof(prm));
  for (int i = 0; i < v->nb; ++i) {
    prm.src[i] = src->blk[i];
    prm.dst[i] = dst->blk[i];
  }
  prm.blocks = v->d_blocks;
  prm.off = v->d_off;
  prm.m = v->m; prm.mpad = v->mpad; prm.kpad = v->kpad;
  prm.p = c->p;
  prm.omega = omega; prm.accumulate = accumulate;
  (void)hipGetLastError();
  if (v->flat) {
    prm.cell = v->d_cell_all;
    prm.cls = v->d_cls_all;
    prm.nquad = v->nquad_all;
    prm.flat = v->d_flat;
    const int rc = vanka_launch(v, prm, prm.nquad, st);
    if (rc != STFEM_OK) {
      snprintf(g_vanka_err, sizeof(g_vanka_err), "vanka_apply_kernel: %s", hipGetErrorString(hipGetLastError()));
      return rc;
    }
    return vanka_collect(v, dst, omega, accumulate, st);
  }
  for (int colour = 0; colour < 8; ++colour) {
    if (v->nquad[colour] == 0) continue;
    prm.cell = v->d_cell[colour];
    prm.cls = v->d_cls[colour];
    prm.nquad = v->nquad[colour];
    prm.colour = colour;
    const int rc = vanka_launch(v, prm, prm.nquad, st);
    if (rc != STFEM_OK) {
      snprintf(g_vanka_err, sizeof(g_vanka_err), "vanka_apply_kernel: %s", hipGetErrorString(hipGetLastError()));
      return rc;
    }
  }
  return STFEM_OK;
}

} // extern "C"
