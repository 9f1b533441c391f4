// Cell-patch Vanka smoother of the two-variable Stokes space-time system (SURVEY 8 f-1 for BASELINE configs[4]).
//
// Replaces PreconditionVanka in its block form (reference include/stmg.h:626-738: the constructor over BlockSparseMatrixType with
// a BlockSlice, K_mask / M_mask; vmult 832-872) as tests/tp_03stokes.cc:537-540, 714-726 sets it up: per cell the block
//     B_c((i, k), (j, l)) = valence_iv(k) * (Alpha(i, j) K_{iv,jv}(k, l) + [iv = jv = 0] Beta(i, j) M(k, l)),
// i, j = blocks of the BlockSlice (time step, variable, time dof), k, l = the cell's DoFs of the block's variable (81 velocity
// DoFs, 8 FE_Q(1) or 4 FE_DGP(1) pressure DoFs), K = the ASSEMBLED Stokes matrix [[nu K, -B^T], [B, 0]] (+ the Nitsche terms of
// weak boundary faces) and M = the assembled vector mass, both restricted to the cell (compute_block_matrix.h:50-139) with the
// strong velocity constraints (row and column dropped, diagonal kept), inverted by Gauss-Jordan;
//     vmult: dst = sum over cells of scatter(B_c^-1 gather(src)).
// Axis-aligned uniform meshes only (the Kronecker path of csrc/stfem_stokes.hip): there the block of a cell depends only on which
// neighbours it has - at most 27 blocks per mesh.  Set-up: every class block is read off the device operator itself, applied to
// unit vectors on a mesh of 1 - 3 cells per direction with the cell in the position of its class (the reference's own method of
// getting matrix entries, tests/tp_05dgp_support.cc:140-149) - no second implementation of the cell matrices.  Apply: the
// MFMA class kernel of the scalar smoother (stfem_vanka_kernel.h) with a row table in place of its (block, node) arithmetic, rows
// to a scratch array, then one collecting launch that sums every DoF's cells in a fixed order: two launches, no colours, no
// atomics, bitwise reproducible.
#include "stfem_internal.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <vector>

#include "stfem_vanka_kernel.h"

namespace {

thread_local char g_sv_err[256] = "";

struct StokesCollectParams {
  double *dst[VK_MAX_BLOCKS];
  const double *y;   // [slot][mpad]
  const int *slot;   // cell -> slot
  int nblk, mpad, pdg;
  int var[VK_MAX_BLOCKS], rowbase[VK_MAX_BLOCKS];
  int nc[3], ndu[3], ndp[3];
  long long Nu, Np;
  double omega;
  int accumulate;
};

// blockIdx.y = block of the BlockSlice; a thread takes one DoF of it: (component, node) of a velocity block, a node or a cell
// function of a pressure block, and sums the rows its cells left in the scratch array (cells in z, y, x order)
__global__ __launch_bounds__(256) void stokes_vanka_collect_kernel(const StokesCollectParams P)
{
  const int b = blockIdx.y;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int var = P.var[b];
  const long long n_dofs = var == 0 ? 3 * P.Nu : P.Np;
  if (i >= n_dofs) return;
  double s = 0.0;
  if (var == 1 && P.pdg) {
    const long long cell = i >> 2;
    s = P.y[size_t(P.slot[cell]) * P.mpad + P.rowbase[b] + int(i & 3)];
  } else {
    const int p = var == 0 ? 2 : 1, np = p + 1;
    const int *nd = var == 0 ? P.ndu : P.ndp;
    const long long N = var == 0 ? P.Nu : P.Np;
    const int comp = int(i / N);
    const long long node = i - (long long)comp * N;
    const int idx[3] = {int(node % nd[0]), int((node / nd[0]) % nd[1]), int(node / ((long long)nd[0] * nd[1]))};
    int cc[3][2], ll[3][2], cnt[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int c0 = idx[d] / p, l0 = idx[d] - c0 * p;
      cnt[d] = 0;
      if (l0 == 0) {
        if (c0 > 0) { cc[d][cnt[d]] = c0 - 1; ll[d][cnt[d]] = p; ++cnt[d]; }
        if (c0 < P.nc[d]) { cc[d][cnt[d]] = c0; ll[d][cnt[d]] = 0; ++cnt[d]; }
      } else {
        cc[d][0] = c0; ll[d][0] = l0; cnt[d] = 1;
      }
    }
    const int rb = P.rowbase[b] + comp * 27;
    for (int kz = 0; kz < cnt[2]; ++kz)
      for (int ky = 0; ky < cnt[1]; ++ky)
        for (int kx = 0; kx < cnt[0]; ++kx) {
          const int cell = cc[0][kx] + P.nc[0] * (cc[1][ky] + P.nc[1] * cc[2][kz]);
          const int n = ll[0][kx] + np * (ll[1][ky] + np * ll[2][kz]);
          s += P.y[size_t(P.slot[cell]) * P.mpad + rb + n];
        }
  }
  double *d = P.dst[b] + i;
  *d = P.accumulate ? *d + P.omega * s : P.omega * s;
}

bool invert_dense(int n, std::vector<double> &a) // Gauss-Jordan with partial pivoting (FullMatrix::gauss_jordan), in place
{
  std::vector<int> piv(n);
  for (int k = 0; k < n; ++k) {
    int r = k;
    double best = std::fabs(a[size_t(k) * n + k]);
    for (int i = k + 1; i < n; ++i)
      if (std::fabs(a[size_t(i) * n + k]) > best) { best = std::fabs(a[size_t(i) * n + k]); r = i; }
    if (!(best > 0.0)) return false;
    piv[k] = r;
    if (r != k)
      for (int j = 0; j < n; ++j) std::swap(a[size_t(k) * n + j], a[size_t(r) * n + j]);
    const double d = 1.0 / a[size_t(k) * n + k];
    a[size_t(k) * n + k] = 1.0;
    for (int j = 0; j < n; ++j) a[size_t(k) * n + j] *= d;
    for (int i = 0; i < n; ++i)
      if (i != k) {
        const double f = a[size_t(i) * n + k];
        if (f == 0.0) continue;
        a[size_t(i) * n + k] = 0.0;
        for (int j = 0; j < n; ++j) a[size_t(i) * n + j] -= f * a[size_t(k) * n + j];
      }
  }
  for (int k = n - 1; k >= 0; --k)
    if (piv[k] != k)
      for (int i = 0; i < n; ++i) std::swap(a[size_t(i) * n + k], a[size_t(i) * n + piv[k]]);
  return true;
}

template <int MT> const void *sv_kernel() { return reinterpret_cast<const void *>(&vanka_apply_kernel<double, 27, MT>); }
const void *sv_kernel(int mtw)
{
  switch (mtw) {
    case 1: return sv_kernel<1>();
    case 2: return sv_kernel<2>();
    case 3: return sv_kernel<3>();
    case 4: return sv_kernel<4>();
    case 6: return sv_kernel<6>();
    default: return nullptr;
  }
}

} // namespace

struct stfem_stokes_vanka {
  stfem_stokes_ctx *ctx = nullptr;
  stfem_stokes_desc d;
  int nblk = 0, var[VK_MAX_BLOCKS] = {0}, rowbase[VK_MAX_BLOCKS] = {0};
  int m = 0, mt = 0, mtw = 0, parts = 0, mpad = 0, kpad = 0, nclasses = 0, nquad = 0;
  double *d_blocks = nullptr, *d_flat = nullptr;
  int2 *d_rowtab = nullptr;
  int *d_cellu = nullptr, *d_cellp = nullptr, *d_cls = nullptr, *d_slot = nullptr;
};

#define SV_TRY(call)                                                                \
  do {                                                                              \
    hipError_t e_ = (call);                                                         \
    if (e_ != hipSuccess) {                                                         \
      snprintf(g_sv_err, sizeof(g_sv_err), "%s: %s", #call, hipGetErrorString(e_)); \
      return STFEM_ERR_HIP;                                                         \
    }                                                                               \
  } while (0)

namespace {

// The restricted assembled matrices of one block class, read off the operator on a mesh of 1 - 3 cells per direction:
// A = [[nu K, -B^T], [B, 0]] (+ weak faces), Mu = vector mass, both (81 + npl)^2 / 81^2 over the cell's DoFs, unconstrained.
int probe_class(const stfem_stokes_desc &d, int key, int npl, std::vector<double> &A, std::vector<double> &Mu)
{
  const int nl = 81 + npl;
  A.assign(size_t(nl) * nl, 0.0);
  Mu.assign(size_t(81) * 81, 0.0);
  stfem_mesh_desc md;
  std::memset(&md, 0, sizeof(md));
  int cc[3], weak = 0;
  for (int k = 0; k < 3; ++k) {
    const int lo = (key >> (2 * k)) & 1, hi = (key >> (2 * k + 1)) & 1;
    md.ncell[k] = 1 + lo + hi;
    const double h = (d.upper[k] - d.lower[k]) / d.nc[k];
    md.lower[k] = 0.0;
    md.upper[k] = h * md.ncell[k];
    cc[k] = lo;
    if (!lo && (d.weak_mask & (1 << (2 * k)))) weak |= 1 << (2 * k);
    if (!hi && (d.weak_mask & (2 << (2 * k)))) weak |= 2 << (2 * k);
  }
  md.vertices = nullptr;
  md.dirichlet_mask = 0; // (the strong constraints are applied to the block afterwards: the diagonal of the unconstrained assembly stays)
  md.device = d.device;
  stfem_stokes_ctx *t = nullptr;
  int rc = stfem_stokes_create_ex(&md, 2, d.pspace, d.nu, &t);
  if (rc != STFEM_OK) return rc;
  if (weak) rc = stfem_stokes_set_weak_boundaries(t, weak, 0, d.penalty1, d.penalty2);
  const long long Nu = stfem_stokes_n_velocity_dofs(t), Np = stfem_stokes_n_pressure_dofs(t);
  const int ndu[3] = {2 * md.ncell[0] + 1, 2 * md.ncell[1] + 1, 2 * md.ncell[2] + 1};
  const int ndp[3] = {md.ncell[0] + 1, md.ncell[1] + 1, md.ncell[2] + 1};
  // global indices (within the u / p vector of the small mesh) of the cell's local DoFs
  std::vector<long long> gi(nl);
  for (int c = 0; c < 3; ++c)
    for (int n = 0; n < 27; ++n) {
      const int a = n % 3, b = (n / 3) % 3, e = n / 9;
      gi[c * 27 + n] = c * Nu + (2 * cc[0] + a) + (long long)ndu[0] * ((2 * cc[1] + b) + (long long)ndu[1] * (2 * cc[2] + e));
    }
  for (int n = 0; n < npl; ++n) {
    if (d.pspace) gi[81 + n] = 4ll * (cc[0] + md.ncell[0] * (cc[1] + md.ncell[1] * cc[2])) + n;
    else {
      const int a = n % 2, b = (n / 2) % 2, e = n / 4;
      gi[81 + n] = (cc[0] + a) + (long long)ndp[0] * ((cc[1] + b) + (long long)ndp[1] * (cc[2] + e));
    }
  }
  double *su = nullptr, *sp = nullptr, *du = nullptr, *dp = nullptr;
  if (rc == STFEM_OK) rc = stfem_stokes_vector_create(t, 0, &su);
  if (rc == STFEM_OK) rc = stfem_stokes_vector_create(t, 1, &sp);
  if (rc == STFEM_OK) rc = stfem_stokes_vector_create(t, 0, &du);
  if (rc == STFEM_OK) rc = stfem_stokes_vector_create(t, 1, &dp);
  const size_t lu = size_t(3 * Nu), lp = size_t(Np);
  std::vector<double> hu(lu, 0.0), hp(lp, 0.0), zu(lu, 0.0), zp(lp, 0.0);
  for (int col = 0; col < nl && rc == STFEM_OK; ++col) {
    std::vector<double> &z = col < 81 ? zu : zp;
    z[size_t(gi[col])] = 1.0;
    rc = stfem_stokes_vector_upload(t, col < 81 ? 0 : 1, col < 81 ? su : sp, z.data());
    if (rc == STFEM_OK) rc = stfem_stokes_vmult(t, du, dp, su, sp, nullptr);
    if (rc == STFEM_OK) rc = stfem_stokes_vector_download(t, 0, du, hu.data());
    if (rc == STFEM_OK) rc = stfem_stokes_vector_download(t, 1, dp, hp.data());
    for (int row = 0; row < nl; ++row) A[size_t(row) * nl + col] = row < 81 ? hu[size_t(gi[row])] : hp[size_t(gi[row])];
    if (rc == STFEM_OK && col < 81) {
      rc = stfem_stokes_mass_vmult(t, du, su, nullptr);
      if (rc == STFEM_OK) rc = stfem_stokes_vector_download(t, 0, du, hu.data());
      for (int row = 0; row < 81; ++row) Mu[size_t(row) * 81 + col] = hu[size_t(gi[row])];
    }
    z[size_t(gi[col])] = 0.0;
    if (rc == STFEM_OK) rc = stfem_stokes_vector_upload(t, col < 81 ? 0 : 1, col < 81 ? su : sp, z.data());
  }
  if (su) stfem_stokes_vector_destroy(t, su);
  if (sp) stfem_stokes_vector_destroy(t, sp);
  if (du) stfem_stokes_vector_destroy(t, du);
  if (dp) stfem_stokes_vector_destroy(t, dp);
  stfem_stokes_destroy(t);
  return rc;
}

} // namespace

extern "C" {

const char *stfem_stokes_vanka_last_error(void) { return g_sv_err; }

void stfem_stokes_vanka_destroy(stfem_stokes_vanka *v)
{
  if (!v) return;
  (void)hipSetDevice(v->d.device);
  if (v->d_blocks) (void)hipFree(v->d_blocks);
  if (v->d_flat) (void)hipFree(v->d_flat);
  if (v->d_rowtab) (void)hipFree(v->d_rowtab);
  if (v->d_cellu) (void)hipFree(v->d_cellu);
  if (v->d_cellp) (void)hipFree(v->d_cellp);
  if (v->d_cls) (void)hipFree(v->d_cls);
  if (v->d_slot) (void)hipFree(v->d_slot);
  delete v;
}

int stfem_stokes_vanka_n_classes(const stfem_stokes_vanka *v) { return v ? v->nclasses : 0; }

int stfem_stokes_vanka_create(stfem_stokes_ctx *ctx, int n_blocks, const int32_t *block_variable, const double *Alpha, const double *Beta,
                              stfem_stokes_vanka **out)
{
  if (!ctx || !block_variable || !Alpha || !Beta || !out || n_blocks < 1) return STFEM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (n_blocks > VK_MAX_BLOCKS) return STFEM_ERR_UNSUPPORTED;
  stfem_stokes_vanka *v = new (std::nothrow) stfem_stokes_vanka;
  if (!v) return STFEM_ERR_OUT_OF_MEMORY;
  v->ctx = ctx;
  int rc = stfem_stokes_internal_desc(ctx, &v->d);
  const stfem_stokes_desc &d = v->d;
  if (rc == STFEM_OK && !d.cart) rc = STFEM_ERR_UNSUPPORTED; // (one block per cell on general meshes: not built for two variables)
  if (rc != STFEM_OK) { delete v; return rc; }
  const int npl = d.pspace ? 4 : 8, nl = 81 + npl;
  v->nblk = n_blocks;
  for (int i = 0; i < n_blocks; ++i) {
    if (block_variable[i] < 0 || block_variable[i] > 1) { delete v; return STFEM_ERR_INVALID_ARGUMENT; }
    v->var[i] = block_variable[i];
    v->rowbase[i] = v->m;
    v->m += block_variable[i] == 0 ? 81 : npl;
  }
  const int m = v->m;
  if (m > VK_MAX_ROWS) { delete v; return STFEM_ERR_UNSUPPORTED; }
  { // row tiles per workgroup: the split with the fewest padded tiles (fp64: at most six per workgroup)
    const int tiles = (m + 15) / 16;
    int best = 1 << 30;
    for (int mtw : {4, 3, 6, 2, 1}) {
      if (mtw > tiles && mtw != 1) continue;
      const int parts = (tiles + mtw - 1) / mtw;
      if (parts * mtw < best) { best = parts * mtw; v->mtw = mtw; v->parts = parts; }
    }
    v->mt = v->parts * v->mtw;
    v->mpad = 16 * v->mt;
    v->kpad = ((m + KS - 1) / KS) * KS;
  }
  SV_TRY(hipSetDevice(d.device));
  // ---- classes: per direction bit 0 = has a lower neighbour, bit 1 = has an upper neighbour
  auto dir_class = [&](int k, int c) { return (c > 0 ? 1 : 0) | (c < d.nc[k] - 1 ? 2 : 0); };
  std::map<int, int> class_id;
  std::vector<int> class_key;
  for (int cz = 0; cz < d.nc[2]; ++cz)
    for (int cy = 0; cy < d.nc[1]; ++cy)
      for (int cx = 0; cx < d.nc[0]; ++cx) {
        const int key = dir_class(0, cx) | (dir_class(1, cy) << 2) | (dir_class(2, cz) << 4);
        if (!class_id.count(key)) { class_id[key] = int(class_key.size()); class_key.push_back(key); }
      }
  v->nclasses = int(class_key.size());
  const size_t bsz = size_t(v->kpad) * v->mpad;
  std::vector<double> all(bsz * v->nclasses, 0.0), A, Mu;
  for (int ci = 0; ci < v->nclasses; ++ci) {
    const int key = class_key[ci];
    rc = probe_class(d, key, npl, A, Mu);
    if (rc != STFEM_OK) {
      snprintf(g_sv_err, sizeof(g_sv_err), "probing the block of class %d: status %d (%s)", key, rc, stfem_stokes_last_hip_error());
      stfem_stokes_vanka_destroy(v);
      return rc;
    }
    // valence and strong constraints of the cell's DoFs
    std::vector<double> val(nl, 1.0);
    std::vector<char> con(nl, 0);
    for (int c = 0; c < 3; ++c)
      for (int n = 0; n < 27; ++n) {
        const int a[3] = {n % 3, (n / 3) % 3, n / 9};
        for (int k = 0; k < 3; ++k) {
          const int kk = (key >> (2 * k)) & 3;
          if ((a[k] == 0 && (kk & 1)) || (a[k] == 2 && (kk & 2))) val[c * 27 + n] *= 2.0;
          if ((a[k] == 0 && !(kk & 1) && (d.dmask & (1 << (2 * k)))) || (a[k] == 2 && !(kk & 2) && (d.dmask & (2 << (2 * k))))) con[c * 27 + n] = 1;
        }
      }
    if (!d.pspace)
      for (int n = 0; n < 8; ++n) {
        const int a[3] = {n % 2, (n / 2) % 2, n / 4};
        for (int k = 0; k < 3; ++k) {
          const int kk = (key >> (2 * k)) & 3;
          if ((a[k] == 0 && (kk & 1)) || (a[k] == 1 && (kk & 2))) val[81 + n] *= 2.0;
        }
      }
    for (int r = 0; r < 81; ++r)
      if (con[r])
        for (int s = 0; s < nl; ++s)
          if (s != r) {
            A[size_t(r) * nl + s] = A[size_t(s) * nl + r] = 0.0;
            if (s < 81) Mu[size_t(r) * 81 + s] = Mu[size_t(s) * 81 + r] = 0.0;
          }
    std::vector<double> B(size_t(m) * m, 0.0);
    for (int i = 0; i < n_blocks; ++i)
      for (int j = 0; j < n_blocks; ++j) {
        const int iv = v->var[i], jv = v->var[j];
        const int ni = iv ? npl : 81, nj = jv ? npl : 81, ro = iv ? 81 : 0, co = jv ? 81 : 0;
        const double al = Alpha[i * n_blocks + j], be = Beta[i * n_blocks + j];
        for (int k = 0; k < ni; ++k)
          for (int l = 0; l < nj; ++l) {
            double e = 0.0;
            if (be != 0.0 && iv == 0 && jv == 0) e += be * Mu[size_t(k) * 81 + l];                 // M_mask(0, 0) only
            if (al != 0.0) e += al * A[size_t(ro + k) * nl + co + l];
            B[size_t(v->rowbase[i] + k) * m + v->rowbase[j] + l] = val[ro + k] * e;
          }
      }
    if (!invert_dense(m, B)) {
      snprintf(g_sv_err, sizeof(g_sv_err), "singular cell block (class %d)", key);
      stfem_stokes_vanka_destroy(v);
      return STFEM_ERR_INVALID_ARGUMENT;
    }
    double *dstb = all.data() + bsz * ci;
    for (int r = 0; r < m; ++r)
      for (int k = 0; k < m; ++k) dstb[size_t(k) * v->mpad + r] = B[size_t(r) * m + k];
  }
  // ---- row table: row -> (vector, variable, element offset from the cell's first DoF of the variable)
  std::vector<int2> rowtab(m);
  for (int i = 0; i < n_blocks; ++i) {
    if (v->var[i] == 0) {
      for (int c = 0; c < 3; ++c)
        for (int n = 0; n < 27; ++n) {
          const int a = n % 3, b = (n / 3) % 3, e = n / 9;
          const long long off = c * d.Nu + a + (long long)d.ndu[0] * (b + (long long)d.ndu[1] * e);
          if (off > 0x7fffffffll) { stfem_stokes_vanka_destroy(v); return STFEM_ERR_UNSUPPORTED; }
          rowtab[v->rowbase[i] + c * 27 + n] = make_int2(i, int(off));
        }
    } else {
      for (int n = 0; n < npl; ++n) {
        const int a = n % 2, b = (n / 2) % 2, e = n / 4;
        rowtab[v->rowbase[i] + n] = make_int2(i | (1 << 8), d.pspace ? n : a + d.ndp[0] * (b + d.ndp[1] * e));
      }
    }
  }
  // ---- cells grouped by class into batches of 16, four batches of one class per workgroup
  const long long ncells = (long long)d.nc[0] * d.nc[1] * d.nc[2];
  std::map<int, std::vector<int>> by_class;
  for (int cz = 0; cz < d.nc[2]; ++cz)
    for (int cy = 0; cy < d.nc[1]; ++cy)
      for (int cx = 0; cx < d.nc[0]; ++cx)
        by_class[class_id[dir_class(0, cx) | (dir_class(1, cy) << 2) | (dir_class(2, cz) << 4)]].push_back(cx + d.nc[0] * (cy + d.nc[1] * cz));
  std::vector<int> cellu, cellp, cls, slot(size_t(ncells), 0);
  for (auto &kv : by_class) {
    for (int cell : kv.second) {
      const int cx = cell % d.nc[0], cy = (cell / d.nc[0]) % d.nc[1], cz = cell / (d.nc[0] * d.nc[1]);
      slot[cell] = int(cellu.size());
      cellu.push_back(2 * cx + d.ndu[0] * (2 * cy + d.ndu[1] * 2 * cz));
      cellp.push_back(d.pspace ? 4 * cell : cx + d.ndp[0] * (cy + d.ndp[1] * cz));
    }
    cellu.resize(((cellu.size() + 63) / 64) * 64, -1);
    cellp.resize(cellu.size(), 0);
    while (cls.size() < cellu.size() / 64) cls.push_back(kv.first);
  }
  v->nquad = int(cls.size());
  if (hipMalloc(&v->d_blocks, all.size() * sizeof(double)) != hipSuccess || hipMalloc(&v->d_rowtab, rowtab.size() * sizeof(int2)) != hipSuccess ||
      hipMalloc(&v->d_cellu, cellu.size() * sizeof(int)) != hipSuccess || hipMalloc(&v->d_cellp, cellp.size() * sizeof(int)) != hipSuccess ||
      hipMalloc(&v->d_cls, cls.size() * sizeof(int)) != hipSuccess || hipMalloc(&v->d_slot, slot.size() * sizeof(int)) != hipSuccess ||
      hipMalloc(&v->d_flat, cellu.size() * v->mpad * sizeof(double)) != hipSuccess) {
    stfem_stokes_vanka_destroy(v);
    return STFEM_ERR_OUT_OF_MEMORY;
  }
  if (hipMemcpy(v->d_blocks, all.data(), all.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(v->d_rowtab, rowtab.data(), rowtab.size() * sizeof(int2), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(v->d_cellu, cellu.data(), cellu.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(v->d_cellp, cellp.data(), cellp.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(v->d_cls, cls.data(), cls.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(v->d_slot, slot.data(), slot.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
    stfem_stokes_vanka_destroy(v);
    return STFEM_ERR_HIP;
  }
  *out = v;
  return STFEM_OK;
}

// dst = (accumulate ? dst : 0) + omega * (sum over cells of scatter(B_c^-1 gather(src))); blocks in the order of the BlockSlice
// the smoother was created with (velocity blocks: 3 n_velocity_dofs doubles, component-major; pressure blocks: n_pressure_dofs)
int stfem_stokes_vanka_step(stfem_stokes_vanka *v, double *const *dst_blocks, double omega, int accumulate, const double *const *src_blocks,
                            void *stream)
{
  if (!v || !dst_blocks || !src_blocks) return STFEM_ERR_INVALID_ARGUMENT;
  for (int i = 0; i < v->nblk; ++i) {
    if (!dst_blocks[i] || !src_blocks[i]) return STFEM_ERR_INVALID_ARGUMENT;
    for (int j = 0; j < v->nblk; ++j)
      if (dst_blocks[i] == src_blocks[j]) return STFEM_ERR_ALIAS;
  }
  struct Scope { // the reference's TimerOutput scope "vanka" (stmg.h:835)
    Scope() { stfem_trace_push("vanka"); }
    ~Scope() { stfem_trace_pop(); }
  } scope;
  SV_TRY(hipSetDevice(v->d.device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  VankaParams prm;
  std::memset(&prm, 0, sizeof(prm));
  for (int i = 0; i < v->nblk; ++i) {
    prm.src[i] = src_blocks[i];
    prm.dst[i] = dst_blocks[i];
  }
  prm.blocks = v->d_blocks;
  prm.cell = v->d_cellu; prm.cell2 = v->d_cellp; prm.cls = v->d_cls; prm.rowtab = v->d_rowtab;
  prm.nquad = v->nquad; prm.m = v->m; prm.mpad = v->mpad; prm.kpad = v->kpad; prm.p = 2;
  prm.flat = v->d_flat; prm.omega = 1.0;
  (void)hipGetLastError();
  const void *k = sv_kernel(v->mtw);
  if (!k) return STFEM_ERR_UNSUPPORTED;
  void *args[] = {&prm};
  if (hipLaunchKernel(k, dim3(v->nquad, v->parts), dim3(256), args, 0, st) != hipSuccess) {
    snprintf(g_sv_err, sizeof(g_sv_err), "vanka_apply_kernel: %s", hipGetErrorString(hipGetLastError()));
    return STFEM_ERR_HIP;
  }
  StokesCollectParams cp;
  std::memset(&cp, 0, sizeof(cp));
  for (int i = 0; i < v->nblk; ++i) { cp.dst[i] = dst_blocks[i]; cp.var[i] = v->var[i]; cp.rowbase[i] = v->rowbase[i]; }
  cp.y = v->d_flat; cp.slot = v->d_slot; cp.nblk = v->nblk; cp.mpad = v->mpad; cp.pdg = v->d.pspace;
  for (int k3 = 0; k3 < 3; ++k3) { cp.nc[k3] = v->d.nc[k3]; cp.ndu[k3] = v->d.ndu[k3]; cp.ndp[k3] = v->d.ndp[k3]; }
  cp.Nu = v->d.Nu; cp.Np = v->d.Np; cp.omega = omega; cp.accumulate = accumulate;
  const long long big = std::max(3 * v->d.Nu, v->d.Np);
  hipLaunchKernelGGL(stokes_vanka_collect_kernel, dim3((unsigned)((big + 255) / 256), v->nblk), dim3(256), 0, st, cp);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_sv_err, sizeof(g_sv_err), "stokes_vanka_collect_kernel: %s", hipGetErrorString(e));
    return STFEM_ERR_HIP;
  }
  return STFEM_OK;
}

int stfem_stokes_vanka_vmult(stfem_stokes_vanka *v, double *const *dst_blocks, const double *const *src_blocks, void *stream)
{
  return stfem_stokes_vanka_step(v, dst_blocks, 1.0, 0, src_blocks, stream);
}

} // extern "C"
