// C-ABI of the MI355X space-time operator apply (see include/stfem.h).
#include "../../include/stfem.h"

#include "host_tables.h"
#include "stfem_internal.h"
#include "stfem_kernels.h"

#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <vector>

using namespace stfem;

// Precision traits: the same host logic drives the fp64 (stfem::f64) and fp32 (stfem::f32)
// instantiations of the device code.
struct Prec64 {
  using real = double;
  using Sweep = f64::SweepParams;
  using Plan = f64::TilePlan;
  using Diag = f64::DiagParams;
  static int atomic(int p, const Sweep &s, void *st) { return f64::launch_cart_atomic(p, s, st); }
  static int geometry(int p, int nbm, int general, Plan &pl) { return f64::tile_geometry(p, nbm, general, pl); }
  static int occupancy(int p, int nbm, int general)
  {
    static int cache[6][stfem::MAX_BLOCKS + 1][2] = {}; // 0 = not asked yet (one device type per process)
    int &v = cache[p][nbm][general];
    if (v == 0) v = std::max(1, f64::tile_occupancy(p, nbm, general)) + 100;
    return v - 100;
  }
  static int tile(int p, const Sweep &s, const Plan &pl, void *st) { return f64::launch_cart_tile(p, s, pl, st); }
  using PPlan = f64::PencilPlan;
  static int pencil_geometry(int p, int nbm, int ty, PPlan &pl) { return f64::pencil_geometry(p, nbm, ty, pl); }
  static int pencil(int p, const Sweep &s, const PPlan &pl, void *st) { return f64::launch_pencil(p, s, pl, st); }
  static const char *pencil_name() { return "st_sweep_pencil<f64>"; }
  static int diagonal(const Diag &d, void *st) { return f64::launch_diagonal(d, st); }
  static int metric(int p, const int nc[3], const double *v, const double *xq, const double *wq, const real *cl,
                    int ll, const real *cm, int ml, real *m, void *st)
  {
    return f64::launch_build_metric(p, nc, v, xq, wq, cl, ll, cm, ml, m, st);
  }
  static const char *tile_name(bool general) { return general ? "st_sweep_cart_tile<f64, stored metric>" : "st_sweep_cart_tile<f64>"; }
  static const char *atomic_name() { return "st_sweep_cart_atomic<f64>"; }
};
struct Prec32 {
  using real = float;
  using Sweep = f32::SweepParams;
  using Plan = f32::TilePlan;
  using Diag = f32::DiagParams;
  static int atomic(int p, const Sweep &s, void *st) { return f32::launch_cart_atomic(p, s, st); }
  static int geometry(int p, int nbm, int general, Plan &pl) { return f32::tile_geometry(p, nbm, general, pl); }
  static int occupancy(int p, int nbm, int general)
  {
    static int cache[6][stfem::MAX_BLOCKS + 1][2] = {};
    int &v = cache[p][nbm][general];
    if (v == 0) v = std::max(1, f32::tile_occupancy(p, nbm, general)) + 100;
    return v - 100;
  }
  static int tile(int p, const Sweep &s, const Plan &pl, void *st) { return f32::launch_cart_tile(p, s, pl, st); }
  using PPlan = f32::PencilPlan;
  static int pencil_geometry(int p, int nbm, int ty, PPlan &pl) { return f32::pencil_geometry(p, nbm, ty, pl); }
  static int pencil(int p, const Sweep &s, const PPlan &pl, void *st) { return f32::launch_pencil(p, s, pl, st); }
  static const char *pencil_name() { return "st_sweep_pencil<f32>"; }
  static int diagonal(const Diag &d, void *st) { return f32::launch_diagonal(d, st); }
  static int metric(int p, const int nc[3], const double *v, const double *xq, const double *wq, const real *cl,
                    int ll, const real *cm, int ml, real *m, void *st)
  {
    return f32::launch_build_metric(p, nc, v, xq, wq, cl, ll, cm, ml, m, st);
  }
  static const char *tile_name(bool general) { return general ? "st_sweep_cart_tile<f32, stored metric>" : "st_sweep_cart_tile<f32>"; }
  static const char *atomic_name() { return "st_sweep_cart_atomic<f32>"; }
};

namespace {

thread_local std::string g_hip_error;

int hip_fail(hipError_t e, const char *what)
{
  g_hip_error = std::string(what) + ": " + hipGetErrorString(e);
  return STFEM_ERR_HIP;
}
#define HIP_TRY(call)                                  \
  do {                                                 \
    hipError_t e_ = (call);                            \
    if (e_ != hipSuccess) return hip_fail(e_, #call); \
  } while (0)

} // namespace

extern "C" {

const char *stfem_strerror(int s)
{
  switch (s) {
    case STFEM_OK: return "ok";
    case STFEM_ERR_INVALID_ARGUMENT: return "invalid argument";
    case STFEM_ERR_UNSUPPORTED: return "unsupported degree / block count / mesh mode";
    case STFEM_ERR_HIP: return "HIP runtime error";
    case STFEM_ERR_NO_DEVICE: return "no HIP device";
    case STFEM_ERR_SHAPE_MISMATCH: return "block count or size mismatch";
    case STFEM_ERR_ALIAS: return "dst aliases src";
    case STFEM_ERR_OUT_OF_MEMORY: return "out of memory";
    case STFEM_ERR_COMM: return "RCCL error";
    default: return "unknown status";
  }
}

const char *stfem_last_hip_error(void) { return g_hip_error.c_str(); }

int stfem_ctx_create(const stfem_mesh_desc *mesh, const stfem_space_desc *space, stfem_ctx **out)
{
  if (!mesh || !space || !out) return STFEM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (space->degree < 1 || space->degree > 5) return STFEM_ERR_UNSUPPORTED;
  if (space->n_q_points_1d != space->degree + 1 || space->n_components != 1)
    return STFEM_ERR_UNSUPPORTED;
  if (space->precision != 0 && space->precision != 1) return STFEM_ERR_UNSUPPORTED;
  for (int d = 0; d < 3; ++d)
    if (mesh->ncell[d] < 1) return STFEM_ERR_INVALID_ARGUMENT;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return STFEM_ERR_NO_DEVICE;
  if (mesh->device < 0 || mesh->device >= ndev) return STFEM_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(mesh->device));
  int n_cu = 0;
  if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, mesh->device) != hipSuccess) n_cu = 0;

  stfem_ctx *c = new (std::nothrow) stfem_ctx;
  if (!c) return STFEM_ERR_OUT_OF_MEMORY;
  c->p = space->degree;
  c->prec = space->precision;
  c->es = c->prec ? sizeof(float) : sizeof(double);
  c->device = mesh->device;
  c->n_cu = n_cu;
  c->ndofs = c->ncells = 1;
  for (int d = 0; d < 3; ++d) {
    c->nc[d] = mesh->ncell[d];
    c->nd[d] = c->p * c->nc[d] + 1;
    c->ndofs *= c->nd[d];
    c->ncells *= c->nc[d];
  }
  c->dmask = mesh->dirichlet_mask & 63;
  try {
    c->tab = make_shape_tables(c->p);
  } catch (...) {
    delete c;
    return STFEM_ERR_INVALID_ARGUMENT;
  }
  if (!mesh->vertices) {
    c->cartesian = true;
    for (int d = 0; d < 3; ++d) {
      c->lower[d] = mesh->lower[d];
      c->h[d] = (mesh->upper[d] - mesh->lower[d]) / c->nc[d];
      if (!(c->h[d] > 0)) {
        delete c;
        return STFEM_ERR_INVALID_ARGUMENT;
      }
    }
  } else {
    // recognise an axis-aligned uniform box (deal.II compresses such cells as "Cartesian")
    const int64_t nvx = c->nc[0] + 1, nvy = c->nc[1] + 1, nvz = c->nc[2] + 1;
    const int64_t nv = nvx * nvy * nvz;
    const double *v = mesh->vertices;
    double lo[3], up[3];
    for (int d = 0; d < 3; ++d) {
      lo[d] = v[d];
      up[d] = v[3 * (nv - 1) + d];
      c->lower[d] = lo[d];
      c->h[d] = (up[d] - lo[d]) / c->nc[d];
    }
    bool cart = c->h[0] > 0 && c->h[1] > 0 && c->h[2] > 0;
    const double tol = 1e-13 * std::max({std::abs(up[0] - lo[0]), std::abs(up[1] - lo[1]),
                                         std::abs(up[2] - lo[2]), 1e-300});
    for (int64_t k = 0, o = 0; k < nvz && cart; ++k)
      for (int64_t j = 0; j < nvy && cart; ++j)
        for (int64_t i = 0; i < nvx; ++i, ++o) {
          if (std::abs(v[3 * o] - (lo[0] + c->h[0] * i)) > tol ||
              std::abs(v[3 * o + 1] - (lo[1] + c->h[1] * j)) > tol ||
              std::abs(v[3 * o + 2] - (lo[2] + c->h[2] * k)) > tol) {
            cart = false;
            break;
          }
        }
    c->cartesian = cart;
    c->vertices.assign(v, v + 3 * nv);
  }
  if (const char *v = getenv("STFEM_VARIANT")) c->variant = std::string(v) == "atomic" ? 1 : (std::string(v) == "tile" ? 2 : 0);
  auto env_int = [](const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; };
  c->env_tile_lz = env_int("STFEM_TILE_LZ", 0);
  c->env_exp = env_int("STFEM_EXP", 0);
  c->env_stagger = env_int("STFEM_STAGGER", 0);
  c->env_stagger_div = std::max(1, env_int("STFEM_STAGGER_DIV", 256));
  c->env_pencil_ty = env_int("STFEM_PENCIL_TY", 0);
  c->env_pencil_lz = env_int("STFEM_PENCIL_LZ", 0);
  c->env_timeline = getenv("STFEM_TIMELINE");
  if (hipMalloc(&c->d_scratch, sizeof(double) * (256 + 8 * 512)) != hipSuccess) { // reduction results [256] + partials [DOT_VECS][DOT_GRID]
    delete c;
    return STFEM_ERR_OUT_OF_MEMORY;
  }
  *out = c;
  return STFEM_OK;
}

void stfem_ctx_destroy(stfem_ctx *c)
{
  if (!c) return;
  (void)hipSetDevice(c->device);
  for (void *&p : c->d_coef)
    if (p) (void)hipFree(p);
  if (c->d_scratch) (void)hipFree(c->d_scratch);
  if (c->d_halo) (void)hipFree(c->d_halo);
  if (c->d_vertices) (void)hipFree(c->d_vertices);
  if (c->d_rule) (void)hipFree(c->d_rule);
  if (c->d_metric) (void)hipFree(c->d_metric);
  if (c->d_timeline) (void)hipFree(c->d_timeline);
  if (c->d_work) (void)hipFree(c->d_work);
  delete c;
}

int64_t stfem_n_dofs(const stfem_ctx *c) { return c ? c->ndofs : 0; }
int64_t stfem_n_cells(const stfem_ctx *c) { return c ? c->ncells : 0; }
int stfem_n_dofs_1d(const stfem_ctx *c, int32_t nd[3])
{
  if (!c || !nd) return STFEM_ERR_INVALID_ARGUMENT;
  for (int d = 0; d < 3; ++d) nd[d] = c->nd[d];
  return STFEM_OK;
}
int stfem_is_cartesian(const stfem_ctx *c) { return c && c->cartesian ? 1 : 0; }
int stfem_ctx_precision(const stfem_ctx *c) { return c ? c->prec : -1; }
const char *stfem_last_kernel_name(const stfem_ctx *c) { return c ? c->last_kernel : ""; }

int stfem_set_coefficient(stfem_ctx *c, int which, int layout, const double *host)
{
  if (!c || which < 0 || which > 1 || layout < 0 || layout > 2) return STFEM_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  if (c->d_coef[which]) {
    HIP_TRY(hipFree(c->d_coef[which]));
    c->d_coef[which] = nullptr;
  }
  c->coef_layout[which] = 0;
  c->metric_valid = false;
  if (layout == 0) return STFEM_OK;
  if (!host) return STFEM_ERR_INVALID_ARGUMENT;
  const int nq = c->p + 1;
  const size_t n = size_t(c->ncells) * (layout == 2 ? size_t(nq) * nq * nq : 1);
  if (hipMalloc(&c->d_coef[which], n * c->es) != hipSuccess) return STFEM_ERR_OUT_OF_MEMORY;
  if (c->prec) {
    std::vector<float> tmp(host, host + n);
    HIP_TRY(hipMemcpy(c->d_coef[which], tmp.data(), n * sizeof(float), hipMemcpyHostToDevice));
  } else {
    HIP_TRY(hipMemcpy(c->d_coef[which], host, n * sizeof(double), hipMemcpyHostToDevice));
  }
  c->coef_layout[which] = layout;
  return STFEM_OK;
}

// ------------------------------------------------------------------------------------ vectors

int stfem_vector_create(stfem_ctx *c, int nb, stfem_vec **out)
{
  if (!c || !out || nb < 1) return STFEM_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  stfem_vec *v = new (std::nothrow) stfem_vec;
  if (!v) return STFEM_ERR_OUT_OF_MEMORY;
  v->ctx = c;
  v->nb = nb;
  v->owns = true;
  v->blk.assign(nb, nullptr);
  for (int b = 0; b < nb; ++b) {
    if (hipMalloc(&v->blk[b], size_t(c->ndofs) * c->es) != hipSuccess) {
      stfem_vector_destroy(v);
      return STFEM_ERR_OUT_OF_MEMORY;
    }
    if (hipMemset(v->blk[b], 0, size_t(c->ndofs) * c->es) != hipSuccess) {
      stfem_vector_destroy(v);
      return STFEM_ERR_HIP;
    }
  }
  *out = v;
  return STFEM_OK;
}

int stfem_vector_wrap(stfem_ctx *c, int nb, void *const *blocks, stfem_vec **out)
{
  if (!c || !out || nb < 1 || !blocks) return STFEM_ERR_INVALID_ARGUMENT;
  stfem_vec *v = new (std::nothrow) stfem_vec;
  if (!v) return STFEM_ERR_OUT_OF_MEMORY;
  v->ctx = c;
  v->nb = nb;
  v->owns = false;
  for (int b = 0; b < nb; ++b) {
    if (!blocks[b]) {
      delete v;
      return STFEM_ERR_INVALID_ARGUMENT;
    }
    v->blk.push_back(blocks[b]);
  }
  *out = v;
  return STFEM_OK;
}

int stfem_vector_rebind(stfem_vec *v, int nb, void *const *blocks)
{
  if (!v || v->owns || nb < 1 || !blocks) return STFEM_ERR_INVALID_ARGUMENT;
  for (int b = 0; b < nb; ++b)
    if (!blocks[b]) return STFEM_ERR_INVALID_ARGUMENT;
  try {
    v->blk.assign(blocks, blocks + nb); // no allocation while the block count does not grow
  } catch (...) {
    return STFEM_ERR_OUT_OF_MEMORY;
  }
  v->nb = nb;
  return STFEM_OK;
}

void stfem_vector_destroy(stfem_vec *v)
{
  if (!v) return;
  if (v->owns) {
    (void)hipSetDevice(v->ctx->device);
    for (void *p : v->blk)
      if (p) (void)hipFree(p);
  }
  delete v;
}

int stfem_vector_n_blocks(const stfem_vec *v) { return v ? v->nb : 0; }
void *stfem_vector_block(const stfem_vec *v, int b)
{
  return (v && b >= 0 && b < v->nb) ? v->blk[b] : nullptr;
}

int stfem_vector_upload(stfem_vec *v, const double *const *host)
{
  if (!v || !host) return STFEM_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(v->ctx->device));
  const size_t n = size_t(v->ctx->ndofs);
  std::vector<float> tmp(v->ctx->prec ? n : 0);
  for (int b = 0; b < v->nb; ++b) {
    if (v->ctx->prec) { // host side is always double; fp32 contexts convert here
      for (size_t i = 0; i < n; ++i) tmp[i] = float(host[b][i]);
      HIP_TRY(hipMemcpy(v->blk[b], tmp.data(), n * sizeof(float), hipMemcpyHostToDevice));
    } else {
      HIP_TRY(hipMemcpy(v->blk[b], host[b], n * sizeof(double), hipMemcpyHostToDevice));
    }
  }
  return STFEM_OK;
}

int stfem_vector_download(const stfem_vec *v, double *const *host)
{
  if (!v || !host) return STFEM_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(v->ctx->device));
  HIP_TRY(hipDeviceSynchronize());
  const size_t n = size_t(v->ctx->ndofs);
  std::vector<float> tmp(v->ctx->prec ? n : 0);
  for (int b = 0; b < v->nb; ++b) {
    if (v->ctx->prec) {
      HIP_TRY(hipMemcpy(tmp.data(), v->blk[b], n * sizeof(float), hipMemcpyDeviceToHost));
      for (size_t i = 0; i < n; ++i) host[b][i] = tmp[i];
    } else {
      HIP_TRY(hipMemcpy(host[b], v->blk[b], n * sizeof(double), hipMemcpyDeviceToHost));
    }
  }
  return STFEM_OK;
}

// ------------------------------------------------------------------------------------ operator

extern "C++" {
template <class PR> static void fill_common(const stfem_ctx *c, typename PR::Sweep &prm)
{
  using real = typename PR::real;
  std::memset(&prm, 0, sizeof(prm));
  prm.ncx = c->nc[0]; prm.ncy = c->nc[1]; prm.ncz = c->nc[2];
  prm.nx = c->nd[0]; prm.ny = c->nd[1]; prm.nz = c->nd[2];
  prm.ncells = c->ncells;
  prm.dmask = c->dmask;
  prm.vol = real(c->h[0] * c->h[1] * c->h[2]);
  prm.ihx2 = real(1.0 / (c->h[0] * c->h[0]));
  prm.ihy2 = real(1.0 / (c->h[1] * c->h[1]));
  prm.ihz2 = real(1.0 / (c->h[2] * c->h[2]));
  const int ne = eo_size(c->p + 1);
  for (int i = 0; i < ne; ++i) {
    prm.eo_Si[i] = real(c->tab.eo_Si[i]);
    prm.eo_L[i] = real(c->tab.eo_L[i]);
    prm.eo_S[i] = real(c->tab.eo_S[i]);
    prm.eo_Dq[i] = real(c->tab.eo_Dq[i]);
    prm.eo_DqT[i] = real(c->tab.eo_DqT[i]);
  }
  for (int i = 0; i < EO_N; ++i) prm.fd_W[i] = real(c->tab.fd_W[i]);
  for (int i = 0; i < 8; ++i) {
    prm.fd_lx[i] = real(c->tab.fd_lam[i] / (c->h[0] * c->h[0]));
    prm.fd_ly[i] = real(c->tab.fd_lam[i] / (c->h[1] * c->h[1]));
    prm.fd_lz[i] = real(c->tab.fd_lam[i] / (c->h[2] * c->h[2]));
  }
}

// Chooses the z-chunking of the tile variant.  One colour launch has columns x ntc workgroups of
// ceil(ncz / ntc) layers (+ about one layer of start-up) that run in rounds of `slots` resident
// workgroups (`resident` per CU: what the runtime reports for the kernel, 2 if unknown); what
// matters is that the last round is full.  Fewer chunks win ties (smaller z-halo).
template <class PL> static void plan_chunks(const stfem_ctx *c, PL &tp, int nbm, int resident)
{
  tp.ntx = (c->nc[0] + tp.cw - 1) / tp.cw;
  tp.nty = (c->nc[1] + tp.rows - 1) / tp.rows;
  const int ncz = c->nc[2];
  int ntc = 1;
  if (c->env_tile_lz > 0) {
    const int lz = std::max(1, std::min(ncz, c->env_tile_lz));
    ntc = (ncz + lz - 1) / lz;
  } else {
    (void)nbm;
    const int wpc = std::max(2, resident); // (planning one-per-CU kernels in rounds of 256 measured 5 % slower)
    const int64_t slots = int64_t(c->n_cu > 0 ? c->n_cu : 256) * wpc;
    const int64_t columns = int64_t((tp.ntx + 1) / 2) * tp.nty; // of the larger colour
    double best = 1e300;
    for (int n = 1; n <= ncz; ++n) {
      const int64_t rounds = (columns * n + slots - 1) / slots;
      const double cost = double(rounds) * ((ncz + n - 1) / n + 1.0) * (1.0 + 1e-3 * n);
      if (cost < best) { best = cost; ntc = n; }
    }
  }
  tp.ntc = ntc;
  tp.lz = (ncz + ntc - 1) / ntc; // longest chunk
  tp.zp = c->p * tp.lz + 1;
}

// Chooses the decomposition of the pencil variant: pencils of cpw x ty cells, four of them stacked
// in y per workgroup, z-chunks such that the last round of resident workgroups (two per CU) is full.
template <class PL> static void plan_pencil(const stfem_ctx *c, PL &pp)
{
  pp.ntx = (c->nc[0] + pp.cpw - 2) / (pp.cpw - 1); // cpw - 1 owned cells per pencil
  const int cyw = pp.ty * 4;
  pp.ntyw = (c->nc[1] + cyw - 1) / cyw;
  const int ncz = c->nc[2];
  // z-chunks: B layers each (z-halo = 1 / (4 B) of the planes), tapering off towards the end of
  // the tile list: tiles are pulled at run time in the order of their numbers, and all tiles of one
  // size take the same time, so equal chunks would leave the last round of the resident workgroups
  // mostly empty (measured: 400 of 512 slots busy on average with 12 equal chunks on cfg 1)
  int B = c->env_pencil_lz > 0 ? c->env_pencil_lz : 8;
  B = std::max(B, (ncz + 47) / 48); // at most 64 chunks
  int n = 0, r = ncz;
  pp.zb[0] = 0;
  while (r > 0) {
    const int sz = c->env_pencil_lz < 0 ? std::min(std::max(-c->env_pencil_lz, (ncz + 47) / 48), r) // (negative: equal chunks, experiments)
                                        : std::min(B, std::max(1, (r + 2) / 3));
    r -= sz;
    pp.zb[n + 1] = pp.zb[n] + sz;
    if (n == 62 && r > 0) { // (cannot happen for B >= ncz / 48; guards the table)
      pp.zb[n + 2] = ncz;
      n += 2;
      r = 0;
      break;
    }
    ++n;
  }
  pp.ntc = n;
  pp.lz = 0;
  for (int q = 0; q < n; ++q) pp.lz = std::max(pp.lz, pp.zb[q + 1] - pp.zb[q]);
  pp.zp = c->p * pp.lz + 1;
}

// (Re)builds the per-quadrature-point metric of the general path.  The coefficients in force
// (operators.h:1152-1162: a coefficient replaces the scaling) are baked in; the flags record
// which of them were used so that a K-only / M-only apply with a different set rebuilds.
template <class PR> static int ensure_metric(stfem_ctx *c, bool use_lap, bool use_mass, hipStream_t st)
{
  using real = typename PR::real;
  const int n = c->p + 1;
  const size_t nm = size_t(c->ncells) * 8 * n * n * n;
  const int flags = (use_lap ? 1 : 0) | (use_mass ? 2 : 0);
  if (c->metric_valid && c->metric_flags == flags) return STFEM_OK;
  if (!c->d_vertices) {
    std::vector<double> v = c->vertices;
    if (v.empty()) { // Cartesian box given by extents
      v.resize(size_t(c->nc[0] + 1) * (c->nc[1] + 1) * (c->nc[2] + 1) * 3);
      size_t o = 0;
      for (int k = 0; k <= c->nc[2]; ++k)
        for (int j = 0; j <= c->nc[1]; ++j)
          for (int i = 0; i <= c->nc[0]; ++i, ++o) {
            v[3 * o] = c->lower[0] + c->h[0] * i;
            v[3 * o + 1] = c->lower[1] + c->h[1] * j;
            v[3 * o + 2] = c->lower[2] + c->h[2] * k;
          }
    }
    // (both uploads complete before the context sees either pointer: a failure half way must not
    // leave d_vertices set and d_rule missing, the next call would launch with a null rule)
    std::vector<double> rule(c->tab.xq);
    rule.insert(rule.end(), c->tab.wq.begin(), c->tab.wq.end());
    double *dv = nullptr, *dr = nullptr;
    if (hipMalloc(&dv, v.size() * sizeof(double)) != hipSuccess) return STFEM_ERR_OUT_OF_MEMORY;
    if (hipMalloc(&dr, rule.size() * sizeof(double)) != hipSuccess) {
      (void)hipFree(dv);
      return STFEM_ERR_OUT_OF_MEMORY;
    }
    if (hipMemcpy(dv, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dr, rule.data(), rule.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipFree(dv);
      (void)hipFree(dr);
      return hip_fail(hipGetLastError(), "metric table upload");
    }
    c->d_vertices = dv;
    c->d_rule = dr;
  }
  if (!c->d_metric && hipMalloc(&c->d_metric, nm * sizeof(real)) != hipSuccess)
    return STFEM_ERR_OUT_OF_MEMORY;
  const int rc = PR::metric(c->p, c->nc, c->d_vertices, c->d_rule, c->d_rule + n,
                            use_lap ? static_cast<const real *>(c->d_coef[1]) : nullptr, use_lap ? c->coef_layout[1] : 0,
                            use_mass ? static_cast<const real *>(c->d_coef[0]) : nullptr, use_mass ? c->coef_layout[0] : 0,
                            static_cast<real *>(c->d_metric), st);
  if (rc != 0) return hip_fail(hipGetLastError(), "build_metric");
  c->metric_valid = true;
  c->metric_flags = flags;
  return STFEM_OK;
}

// for the other translation units of the library (stfem_internal.h): the stored metric in the record layout,
// built with the coefficient tables in force (what K and M of a SystemMatrix see)
int stfem_internal_metric(stfem_ctx *c, const void **metric, void *stream)
{
  const bool lap = c->coef_layout[1] != 0, mass = c->coef_layout[0] != 0;
  const int rc = c->prec ? ensure_metric<Prec32>(c, lap, mass, static_cast<hipStream_t>(stream))
                         : ensure_metric<Prec64>(c, lap, mass, static_cast<hipStream_t>(stream));
  if (rc == STFEM_OK) *metric = c->d_metric;
  return rc;
}

int stfem_internal_set_gradient(stfem_ctx *c, const double *p, const double (*w)[2][3][2], double scale)
{
  if (!c) return STFEM_ERR_INVALID_ARGUMENT;
  c->grad_p = p;
  c->grad_scale = scale;
  c->grad_applied = false;
  if (p && w) std::memcpy(c->grad_w, w, sizeof(c->grad_w));
  return STFEM_OK;
}

// a(j,i), b(j,i): effective nbo x nbi matrices (row-major)
template <class PR>
static int apply_tiled_t(stfem_ctx *c, int nbo, int nbi, const std::vector<double> &a,
                       const std::vector<double> &b, stfem_vec *dst, const stfem_vec *src, int add,
                       bool use_lap_coef, bool use_mass_coef, void *stream)
{
  using real = typename PR::real;
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int j = 0; j < nbo; ++j)
    for (int i = 0; i < nbi; ++i)
      if (dst->blk[j] == src->blk[i]) return STFEM_ERR_ALIAS;
  // general path: non-Cartesian cells or per-quadrature-point coefficients -> metric terms
  const bool general = !c->cartesian || c->coef_layout[0] == 2 || c->coef_layout[1] == 2;
  if (general) {
    const int rc = ensure_metric<PR>(c, use_lap_coef, use_mass_coef, st);
    if (rc != STFEM_OK) return rc;
  }
  const bool atomic = c->variant == 1 && !general;
  if (!add && atomic)
    for (int j = 0; j < nbo; ++j)
      HIP_TRY(hipMemsetAsync(dst->blk[j], 0, size_t(c->ndofs) * sizeof(real), st));
  typename PR::Sweep prm;
  fill_common<PR>(c, prm);
  prm.coef_lap = (use_lap_coef && !general) ? static_cast<const real *>(c->d_coef[1]) : nullptr;
  prm.coef_mass = (use_mass_coef && !general) ? static_cast<const real *>(c->d_coef[0]) : nullptr;
  if (general) {
    prm.metric = static_cast<const real *>(c->d_metric);
    prm.vol = real(1); // detJ and the weights live in the metric
  }
  prm.experiment = c->env_exp;
  prm.gp = nullptr;
  // Systems with more blocks than one launch takes are cut into panels (dst += for the later column panels).  On the
  // pencil path a launch needs two cells per wave - Q4 with seven or eight blocks has one - so those systems
  // are cut into equal panels of a size the pencil sweep has (Q4 x 8 blocks: 2 x 2 panels of four).
  int panel = MAX_BLOCKS;
  if (c->variant == 0 && !general && c->p <= 4) { // (FE_Q(5) has no pencil sweep: the tile sweep takes up to MAX_BLOCKS blocks)
    const int need = std::min(MAX_BLOCKS, std::max(nbo, nbi));
    typename PR::PPlan probe;
    std::memset(&probe, 0, sizeof(probe));
    if (PR::pencil_geometry(c->p, need, 0, probe) != 0) {
      int maxp = need;
      while (maxp > 1 && PR::pencil_geometry(c->p, maxp, 0, probe) != 0) --maxp;
      const int parts = (need + maxp - 1) / maxp;
      panel = (need + parts - 1) / parts;
    }
  }
  for (int j0 = 0; j0 < nbo; j0 += panel) {
    bool first = true; // first launch into this row panel overwrites dst unless add
    for (int i0 = 0; i0 < nbi; i0 += panel) {
      const int tj = std::min(panel, nbo - j0), ti = std::min(panel, nbi - i0);
      bool nonzero = false;
      for (int j = 0; j < tj; ++j)
        for (int i = 0; i < ti; ++i) {
          prm.alpha[j * ti + i] = real(a[size_t(j0 + j) * nbi + i0 + i]);
          prm.beta[j * ti + i] = real(b[size_t(j0 + j) * nbi + i0 + i]);
          nonzero = nonzero || prm.alpha[j * ti + i] != real(0) || prm.beta[j * ti + i] != real(0);
        }
      // the reference skips exact zeros too (operators.h:551,556); a panel may only be skipped
      // if something else still defines dst
      const bool last_panel = i0 + panel >= nbi;
      if (!nonzero && (atomic || add || !first || !last_panel)) continue;
      prm.nbo = tj;
      prm.nbi = ti;
      for (int j = 0; j < tj; ++j) prm.dst[j] = static_cast<real *>(dst->blk[j0 + j]);
      for (int i = 0; i < ti; ++i) prm.src[i] = static_cast<const real *>(src->blk[i0 + i]);
      int rc;
      typename PR::PPlan pp;
      std::memset(&pp, 0, sizeof(pp));
      const int pencil_ty = c->env_pencil_ty; // 0: the kernel's default for (p, n_blocks)
      if (atomic) {
        rc = PR::atomic(c->p, prm, st);
        c->last_kernel = PR::atomic_name();
      } else if (c->variant == 0 && !general && PR::pencil_geometry(c->p, std::max(tj, ti), pencil_ty, pp) == 0) {
        plan_pencil(c, pp);
        const int nbm = std::max(tj, ti);
        const int nbm_r = nbm <= 4 ? nbm : (nbm <= 6 ? 6 : 8);
        const size_t ntiles = size_t(pp.ntx) * pp.ntyw * pp.ntc;
        const size_t nyh = ntiles * nbm_r * pp.zp * pp.tX, nzh = ntiles * nbm_r * pp.tYW * pp.tX;
        if (nyh + nzh > c->halo_doubles) {
          HIP_TRY(hipStreamSynchronize(st));
          if (c->d_halo) HIP_TRY(hipFree(c->d_halo));
          c->d_halo = nullptr;
          c->halo_doubles = 0;
          if (hipMalloc(&c->d_halo, (nyh + nzh) * sizeof(real)) != hipSuccess) return STFEM_ERR_OUT_OF_MEMORY;
          c->halo_doubles = nyh + nzh;
        }
        pp.yh = static_cast<real *>(c->d_halo);
        pp.zh = pp.yh + nyh;
        pp.add = (add || !first) ? 1 : 0;
        prm.gp = nullptr;
        if constexpr (sizeof(real) == 8) { // the Stokes gradient term rides in this launch (three FE_Q(2) blocks, dst = ..., no coefficient tables)
          if (c->grad_p && c->p == 2 && tj == 3 && ti == 3 && !pp.add && last_panel && !prm.coef_lap && !prm.coef_mass) {
            prm.gp = c->grad_p;
            prm.gscale = c->grad_scale;
            std::memcpy(prm.gw, c->grad_w, sizeof(prm.gw));
            c->grad_applied = true;
          }
        }
        if (!c->d_work) {
          if (hipMalloc(&c->d_work, 8 * 32 * sizeof(int)) != hipSuccess) return STFEM_ERR_OUT_OF_MEMORY;
          HIP_TRY(hipMemsetAsync(c->d_work, 0, 8 * 32 * sizeof(int), st));
        }
        pp.work = c->d_work;
        pp.grid = 2 * (c->n_cu > 0 ? c->n_cu : 256); // two 4-wave workgroups per CU (registers, LDS)
        // diagnostic builds only (tools/build_pencil_exp.sh -DSTFEM_PENCIL_TIMELINE): phase timestamps of
        // the even-colour launch, dumped to the file named by STFEM_TIMELINE after every apply
        const size_t ptl_n = ntiles * 4 * pp.lz * pp.ty * 8;
        if (c->env_timeline) {
          if (c->tl_n < ptl_n) {
            if (c->d_timeline) HIP_TRY(hipFree(c->d_timeline));
            c->d_timeline = nullptr;
            if (hipMalloc(&c->d_timeline, ptl_n * sizeof(long long)) != hipSuccess) return STFEM_ERR_OUT_OF_MEMORY;
            c->tl_n = ptl_n;
          }
          HIP_TRY(hipMemsetAsync(c->d_timeline, 0, ptl_n * sizeof(long long), st));
          pp.timeline = c->d_timeline;
        }
        rc = PR::pencil(c->p, prm, pp, st);
        c->last_kernel = PR::pencil_name();
        if (c->env_timeline && rc == 0) {
          HIP_TRY(hipStreamSynchronize(st));
          std::vector<long long> h(ptl_n);
          HIP_TRY(hipMemcpy(h.data(), c->d_timeline, ptl_n * sizeof(long long), hipMemcpyDeviceToHost));
          if (FILE *f = fopen(c->env_timeline, "wb")) {
            const long long hdr[4] = {(long long)ntiles, 4, (long long)pp.lz * pp.ty, 8};
            fwrite(hdr, sizeof(long long), 4, f);
            fwrite(h.data(), sizeof(long long), ptl_n, f);
            fclose(f);
          }
        }
      } else {
        typename PR::Plan tp;
        std::memset(&tp, 0, sizeof(tp));
        const int nbm = std::max(tj, ti);
        if (PR::geometry(c->p, nbm, general ? 1 : 0, tp) != 0) return STFEM_ERR_UNSUPPORTED;
        plan_chunks(c, tp, nbm, PR::occupancy(c->p, nbm, general ? 1 : 0));
        const int nbm_r = nbm <= 4 ? nbm : (nbm <= 6 ? 6 : 8);
        const size_t ntiles = size_t(tp.ntx) * tp.nty * tp.ntc;
        const size_t nyh = ntiles * nbm_r * tp.zp * tp.tX, nzh = ntiles * nbm_r * tp.tY * tp.tX,
                     nxs = ntiles * nbm_r * tp.zp * tp.tY;
        if (nyh + nzh + 2 * nxs > c->halo_doubles) {
          HIP_TRY(hipStreamSynchronize(st));
          if (c->d_halo) HIP_TRY(hipFree(c->d_halo));
          c->d_halo = nullptr;
          c->halo_doubles = 0;
          if (hipMalloc(&c->d_halo, (nyh + nzh + 2 * nxs) * sizeof(real)) != hipSuccess)
            return STFEM_ERR_OUT_OF_MEMORY;
          c->halo_doubles = nyh + nzh + 2 * nxs;
        }
        tp.yh = static_cast<real *>(c->d_halo);
        tp.zh = tp.yh + nyh;
        tp.xl = tp.zh + nzh;
        tp.xr = tp.xl + nxs;
        tp.add = (add || !first) ? 1 : 0;
        tp.experiment = c->env_exp;
        tp.stagger = c->env_stagger;
        tp.stagger_div = c->env_stagger_div;
        // diagnostic builds only (tools/build_abl.sh -DSTFEM_TIMELINE): phase timestamps of the
        // even-colour launch, dumped to the file named by STFEM_TIMELINE after every apply
        const char *tl_path = c->env_timeline;
        const size_t tl_n = size_t(tp.ntx) * tp.nty * tp.ntc * 4 * tp.wx * tp.lz * 16;
        if (tl_path) {
          if (c->tl_n < tl_n) {
            if (c->d_timeline) HIP_TRY(hipFree(c->d_timeline));
            c->d_timeline = nullptr;
            if (hipMalloc(&c->d_timeline, tl_n * sizeof(long long)) != hipSuccess) return STFEM_ERR_OUT_OF_MEMORY;
            c->tl_n = tl_n;
          }
          HIP_TRY(hipMemsetAsync(c->d_timeline, 0, tl_n * sizeof(long long), st));
          tp.timeline = c->d_timeline;
        }
        long long *tl_dev = c->d_timeline;
        rc = PR::tile(c->p, prm, tp, st);
        c->last_kernel = PR::tile_name(prm.metric != nullptr);
        if (tl_path && rc == 0) {
          HIP_TRY(hipStreamSynchronize(st));
          std::vector<long long> h(tl_n);
          HIP_TRY(hipMemcpy(h.data(), tl_dev, tl_n * sizeof(long long), hipMemcpyDeviceToHost));
          if (FILE *f = fopen(tl_path, "wb")) {
            const long long hdr[4] = {(long long)tp.ntx * tp.nty * tp.ntc, 4 * tp.wx, tp.lz, 16};
            fwrite(hdr, sizeof(long long), 4, f);
            fwrite(h.data(), sizeof(long long), tl_n, f);
            fclose(f);
          }
        }
      }
      if (rc == -3) return hip_fail(hipGetLastError(), "kernel launch");
      if (rc != 0) return STFEM_ERR_UNSUPPORTED;
      first = false;
    }
  }
  return STFEM_OK;
}

} // extern "C++"

static int apply_tiled(stfem_ctx *c, int nbo, int nbi, const std::vector<double> &a, const std::vector<double> &b,
                       stfem_vec *dst, const stfem_vec *src, int add, bool use_lap_coef, bool use_mass_coef,
                       void *stream)
{
  return c->prec ? apply_tiled_t<Prec32>(c, nbo, nbi, a, b, dst, src, add, use_lap_coef, use_mass_coef, stream)
                 : apply_tiled_t<Prec64>(c, nbo, nbi, a, b, dst, src, add, use_lap_coef, use_mass_coef, stream);
}

// ---- named trace ranges (roctx): the reference's TimerOutput scopes "vmult" / "Tvmult" (operators.h:539, 564, 590), "vanka"
// (stmg.h:835), "gmg" (stmg.h:1335, 1352) show up under the same names in `rocprofv3 --marker-trace`.  The roctx library of
// the profiler SDK is bound at run time; without it the calls do nothing.
namespace {
struct Roctx {
  int (*push)(const char *) = nullptr;
  int (*pop)() = nullptr;
};
const Roctx &roctx()
{
  static Roctx r = [] {
    Roctx q;
    if (const char *e = getenv("STFEM_TRACE"))
      if (atoi(e) == 0) return q;
    void *h = nullptr;
    for (const char *n : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
      h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
    if (!h) return q;
    q.push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
    q.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
    if (!q.push || !q.pop) q.push = nullptr, q.pop = nullptr;
    return q;
  }();
  return r;
}
} // namespace
struct TraceScope {
  explicit TraceScope(const char *name) { stfem_trace_push(name); }
  ~TraceScope() { stfem_trace_pop(); }
};
void stfem_trace_push(const char *name)
{
  if (roctx().push) (void)roctx().push(name ? name : "stfem");
}
void stfem_trace_pop(void)
{
  if (roctx().pop) (void)roctx().pop();
}

int stfem_st_vmult(stfem_ctx *c, int nrows, int ncols, const double *alpha, const double *beta,
                   int transpose, int add, stfem_vec *dst, const stfem_vec *src, void *stream)
{
  if (!c || !alpha || !beta || !dst || !src || nrows < 1 || ncols < 1)
    return STFEM_ERR_INVALID_ARGUMENT;
  TraceScope scope(transpose ? "Tvmult" : "vmult");
  if (dst->ctx != c || src->ctx != c) return STFEM_ERR_INVALID_ARGUMENT;
  const int nbi = transpose ? nrows : ncols, nbo = transpose ? ncols : nrows;
  if (src->nb != nbi || dst->nb != nbo) return STFEM_ERR_SHAPE_MISMATCH;
  std::vector<double> a(size_t(nbo) * nbi), b(size_t(nbo) * nbi);
  for (int j = 0; j < nbo; ++j)
    for (int i = 0; i < nbi; ++i) {
      const size_t s = transpose ? size_t(i) * ncols + j : size_t(j) * ncols + i;
      a[size_t(j) * nbi + i] = alpha[s];
      b[size_t(j) * nbi + i] = beta[s];
    }
  // K = MatrixFreeOperator(0,1), M = MatrixFreeOperator(1,0); a coefficient replaces the 1
  return apply_tiled(c, nbo, nbi, a, b, dst, src, add, c->coef_layout[1] != 0,
                     c->coef_layout[0] != 0, stream);
}

int stfem_space_vmult(stfem_ctx *c, double ms, double ls, stfem_vec *dst, const stfem_vec *src,
                      void *stream)
{
  if (!c || !dst || !src) return STFEM_ERR_INVALID_ARGUMENT;
  if (dst->ctx != c || src->ctx != c) return STFEM_ERR_INVALID_ARGUMENT;
  if (dst->nb != 1 || src->nb != 1) return STFEM_ERR_SHAPE_MISMATCH;
  // operators.h:1152-1162: term present iff scaling != 0; coefficient (if any) replaces it
  const bool lc = ls != 0.0 && c->coef_layout[1] != 0, mc = ms != 0.0 && c->coef_layout[0] != 0;
  std::vector<double> a{ls != 0.0 ? (lc ? 1.0 : ls) : 0.0}, b{ms != 0.0 ? (mc ? 1.0 : ms) : 0.0};
  return apply_tiled(c, 1, 1, a, b, dst, src, 0, lc, mc, stream);
}

extern "C++" {
template <class PR> static int diagonal_t(stfem_ctx *c, double ms, double ls, stfem_vec *diag, void *stream)
{
  using real = typename PR::real;
  if (!c || !diag || diag->ctx != c) return STFEM_ERR_INVALID_ARGUMENT;
  if (diag->nb != 1) return STFEM_ERR_SHAPE_MISMATCH;
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool lc = ls != 0.0 && c->coef_layout[1] != 0, mc = ms != 0.0 && c->coef_layout[0] != 0;
  const bool general = !c->cartesian || c->coef_layout[0] == 2 || c->coef_layout[1] == 2;
  if (general) {
    const int rc = ensure_metric<PR>(c, lc, mc, st);
    if (rc != STFEM_OK) return rc;
  }
  HIP_TRY(hipMemsetAsync(diag->blk[0], 0, size_t(c->ndofs) * sizeof(real), st));
  typename PR::Diag prm;
  std::memset(&prm, 0, sizeof(prm));
  prm.diag = static_cast<real *>(diag->blk[0]);
  prm.ncx = c->nc[0]; prm.ncy = c->nc[1]; prm.ncz = c->nc[2];
  prm.nx = c->nd[0]; prm.ny = c->nd[1];
  prm.p = c->p;
  prm.dmask = c->dmask;
  prm.ms = real(ms != 0.0 ? (mc ? 1.0 : ms) : 0.0); // operators.h:1152-1162
  prm.ls = real(ls != 0.0 ? (lc ? 1.0 : ls) : 0.0);
  prm.vol = real(c->h[0] * c->h[1] * c->h[2]);
  prm.ihx2 = real(1.0 / (c->h[0] * c->h[0]));
  prm.ihy2 = real(1.0 / (c->h[1] * c->h[1]));
  prm.ihz2 = real(1.0 / (c->h[2] * c->h[2]));
  const int n = c->p + 1;
  if (general) {
    prm.metric = static_cast<const real *>(c->d_metric);
  } else {
    prm.coef_lap = lc ? static_cast<const real *>(c->d_coef[1]) : nullptr;
    prm.coef_mass = mc ? static_cast<const real *>(c->d_coef[0]) : nullptr;
  }
  for (int a = 0; a < n; ++a) {
    double m = 0, l = 0;
    for (int q = 0; q < n; ++q) {
      m += c->tab.wq[q] * c->tab.S[q * n + a] * c->tab.S[q * n + a];
      l += c->tab.wq[q] * c->tab.D[q * n + a] * c->tab.D[q * n + a];
    }
    prm.m1[a] = real(m);
    prm.l1[a] = real(l);
  }
  for (int i = 0; i < n * n; ++i) {
    prm.S[i] = real(c->tab.S[i]);
    prm.D[i] = real(c->tab.D[i]);
  }
  if (PR::diagonal(prm, st) != 0) return hip_fail(hipGetLastError(), "diagonal launch");
  return STFEM_OK;
}

} // extern "C++"

int stfem_diagonal(stfem_ctx *c, double ms, double ls, stfem_vec *diag, void *stream)
{
  if (!c) return STFEM_ERR_INVALID_ARGUMENT;
  return c->prec ? diagonal_t<Prec32>(c, ms, ls, diag, stream) : diagonal_t<Prec64>(c, ms, ls, diag, stream);
}

extern "C++" {
// d <- |d| > tol ? 1 / d : 1   (operators.h:1107-1109)
template <typename T> __global__ __launch_bounds__(256) void invert_diagonal_kernel(int64_t n, T *d, T tol)
{
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const T v = d[i];
    d[i] = fabs(v) > tol ? T(1) / v : T(1);
  }
}
// out = a * x + b * y
template <typename T>
__global__ __launch_bounds__(256) void lincomb_kernel(int64_t n, T a, const T *x, T b, const T *y, T *out)
{
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x)
    out[i] = a * x[i] + b * y[i];
}
template <typename T> static int invert_diagonal(stfem_ctx *c, void *d, hipStream_t st)
{
  const unsigned grid = (unsigned)std::min<int64_t>((c->ndofs + 255) / 256, 4096);
  hipLaunchKernelGGL(invert_diagonal_kernel<T>, dim3(grid), dim3(256), 0, st, c->ndofs, static_cast<T *>(d),
                     std::sqrt(std::numeric_limits<T>::epsilon()));
  return hipGetLastError() == hipSuccess ? STFEM_OK : STFEM_ERR_HIP;
}
} // extern "C++"

int stfem_diagonal_inverse(stfem_ctx *c, double ms, double ls, stfem_vec *diag, void *stream)
{
  const int rc = stfem_diagonal(c, ms, ls, diag, stream);
  if (rc != STFEM_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  return c->prec ? invert_diagonal<float>(c, diag->blk[0], st) : invert_diagonal<double>(c, diag->blk[0], st);
}

int stfem_st_diagonal(stfem_ctx *c, int n, const double *alpha, const double *beta, int inverse, stfem_vec *diag,
                      void *stream)
{
  if (!c || !alpha || !beta || !diag || n < 1 || diag->ctx != c) return STFEM_ERR_INVALID_ARGUMENT;
  if (diag->nb != n) return STFEM_ERR_SHAPE_MISMATCH;
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  // diag K and diag M (or their guarded inverses) once, then one linear combination per block
  stfem_vec *dk = nullptr, *dm = nullptr;
  int rc = stfem_vector_create(c, 1, &dk);
  if (rc == STFEM_OK) rc = stfem_vector_create(c, 1, &dm);
  if (rc == STFEM_OK) rc = inverse ? stfem_diagonal_inverse(c, 0.0, 1.0, dk, stream) : stfem_diagonal(c, 0.0, 1.0, dk, stream);
  if (rc == STFEM_OK) rc = inverse ? stfem_diagonal_inverse(c, 1.0, 0.0, dm, stream) : stfem_diagonal(c, 1.0, 0.0, dm, stream);
  const unsigned grid = (unsigned)std::min<int64_t>((c->ndofs + 255) / 256, 4096);
  for (int i = 0; i < n && rc == STFEM_OK; ++i) {
    const double a = inverse ? 1.0 / alpha[size_t(i) * n + i] : alpha[size_t(i) * n + i];
    const double b = inverse ? 1.0 / beta[size_t(i) * n + i] : beta[size_t(i) * n + i];
    if (c->prec)
      hipLaunchKernelGGL(lincomb_kernel<float>, dim3(grid), dim3(256), 0, st, c->ndofs, float(a),
                         static_cast<const float *>(dk->blk[0]), float(b), static_cast<const float *>(dm->blk[0]),
                         static_cast<float *>(diag->blk[i]));
    else
      hipLaunchKernelGGL(lincomb_kernel<double>, dim3(grid), dim3(256), 0, st, c->ndofs, a,
                         static_cast<const double *>(dk->blk[0]), b, static_cast<const double *>(dm->blk[0]),
                         static_cast<double *>(diag->blk[i]));
    if (hipGetLastError() != hipSuccess) rc = STFEM_ERR_HIP;
  }
  if (hipStreamSynchronize(st) != hipSuccess && rc == STFEM_OK) rc = STFEM_ERR_HIP; // the temporaries go away below
  stfem_vector_destroy(dk);
  stfem_vector_destroy(dm);
  return rc;
}

// ------------------------------------------------------------------------------------ BLAS-1 / halo

extern "C++" {
template <typename T> struct AxpyArgs {
  const T *x[MAX_BLOCKS];
  T coef[MAX_BLOCKS];
  int n;
};
template <typename T> __global__ __launch_bounds__(256) void axpy_kernel(int64_t n, AxpyArgs<T> a, T *y)
{
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    T acc = y[i];
    for (int t = 0; t < a.n; ++t) acc = fma(a.coef[t], a.x[t][i], acc);
    y[i] = acc;
  }
}

template <typename T>
static int tensorproduct_add_t(stfem_ctx *c, int nrows, int ncols, const double *A, stfem_vec *cv, const stfem_vec *b,
                               hipStream_t st)
{
  for (int i = 0; i < nrows; ++i)
    for (int j0 = 0; j0 < ncols; j0 += MAX_BLOCKS) {
      AxpyArgs<T> a;
      a.n = 0;
      for (int j = j0; j < std::min(ncols, j0 + MAX_BLOCKS); ++j)
        if (A[size_t(i) * ncols + j] != 0.0) { // operators.h:246
          if (cv->blk[i] == b->blk[j]) return STFEM_ERR_ALIAS;
          a.x[a.n] = static_cast<const T *>(b->blk[j]);
          a.coef[a.n++] = T(A[size_t(i) * ncols + j]);
        }
      if (a.n == 0) continue;
      const unsigned grid = (unsigned)std::min<int64_t>((c->ndofs + 255) / 256, 256 * 16);
      hipLaunchKernelGGL(axpy_kernel<T>, dim3(grid), dim3(256), 0, st, c->ndofs, a, static_cast<T *>(cv->blk[i]));
    }
  return STFEM_OK;
}

} // extern "C++"

int stfem_tensorproduct_add(stfem_ctx *c, int nrows, int ncols, const double *A, stfem_vec *cv,
                            const stfem_vec *b, void *stream)
{
  if (!c || !A || !cv || !b) return STFEM_ERR_INVALID_ARGUMENT;
  if (cv->nb != nrows || b->nb != ncols) return STFEM_ERR_SHAPE_MISMATCH;
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  return c->prec ? tensorproduct_add_t<float>(c, nrows, ncols, A, cv, b, st)
                 : tensorproduct_add_t<double>(c, nrows, ncols, A, cv, b, st);
}

extern "C++" {
// Local inner products, accumulated in double for both precisions, in TWO STAGES with a fixed summation order (bitwise
// reproducible: round 2 finished with a device atomic): every workgroup of stage 1 writes its partial sums, one
// workgroup of stage 2 adds them in index order.  One launch pair handles up to DOT_VECS left-hand vectors against the
// same right-hand vector over all spatial blocks (the Gram-Schmidt step of the Krylov solvers: k inner products, one pass
// over w per group of eight, one read-back).
constexpr int DOT_VECS = 8, DOT_GRID = 512;
struct DotArgs {
  const void *a[DOT_VECS][MAX_BLOCKS];
  const void *b[MAX_BLOCKS];
  int nvec, nblk;
};
template <typename T>
__global__ __launch_bounds__(256) void multi_dot_kernel(int64_t n, const DotArgs args, double *partial /* [nvec][gridDim.x] */)
{
  __shared__ double red[DOT_VECS][4];
  double s[DOT_VECS];
#pragma unroll
  for (int v = 0; v < DOT_VECS; ++v) s[v] = 0.0;
  for (int blk = 0; blk < args.nblk; ++blk) {
    const T *b = static_cast<const T *>(args.b[blk]);
    for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
      const double w = double(b[i]);
#pragma unroll
      for (int v = 0; v < DOT_VECS; ++v)
        if (v < args.nvec) s[v] = fma(double(static_cast<const T *>(args.a[v][blk])[i]), w, s[v]);
    }
  }
#pragma unroll
  for (int v = 0; v < DOT_VECS; ++v) {
    for (int off = 32; off > 0; off >>= 1) s[v] += __shfl_down(s[v], off, 64);
    if ((threadIdx.x & 63) == 0) red[v][threadIdx.x >> 6] = s[v];
  }
  __syncthreads();
  if (threadIdx.x < unsigned(args.nvec)) {
    const int v = threadIdx.x;
    partial[v * gridDim.x + blockIdx.x] = (red[v][0] + red[v][1]) + (red[v][2] + red[v][3]);
  }
}
// stage 2: out[v] = sum of the partials of vector v in index order (a tree with fixed shape)
__global__ __launch_bounds__(256) void dot_finish_kernel(int nvec, int nparts, const double *partial, double *out)
{
  __shared__ double red[256];
  for (int v = 0; v < nvec; ++v) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) s += partial[v * nparts + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (int(threadIdx.x) < w) red[threadIdx.x] += red[threadIdx.x + w];
      __syncthreads();
    }
    if (threadIdx.x == 0) out[v] = red[0];
    __syncthreads();
  }
}
struct MultiAxpyArgs {
  const void *x[DOT_VECS][MAX_BLOCKS];
  void *y[MAX_BLOCKS];
  double coef[DOT_VECS];
  const double *dcoef; // coefficients on the device (sign applied below), or nullptr: coef
  double sign;
  int nvec;
};
// y += sign * sum_v coef_v x_v on every spatial block (blockIdx.y)
template <typename T> __global__ __launch_bounds__(256) void multi_axpy_kernel(int64_t n, const MultiAxpyArgs args)
{
  const int blk = blockIdx.y;
  T *y = static_cast<T *>(args.y[blk]);
  double c[DOT_VECS];
#pragma unroll
  for (int v = 0; v < DOT_VECS; ++v) c[v] = v < args.nvec ? args.sign * (args.dcoef ? args.dcoef[v] : args.coef[v]) : 0.0;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    double acc = double(y[i]);
#pragma unroll
    for (int v = 0; v < DOT_VECS; ++v)
      if (v < args.nvec) acc = fma(c[v], double(static_cast<const T *>(args.x[v][blk])[i]), acc);
    y[i] = T(acc);
  }
}

// d_out[0 .. k): <a_i, b> over the first n_own entries of every block; stays on the device
static int multi_dot_device(stfem_ctx *c, int k, const stfem_vec *const *as, const stfem_vec *b, int64_t n_own, double *d_out, hipStream_t st)
{
  if (b->nb > MAX_BLOCKS) return STFEM_ERR_UNSUPPORTED;
  const int grid = (int)std::min<int64_t>((n_own + 255) / 256, DOT_GRID);
  double *partial = c->d_scratch + 256; // [DOT_VECS][DOT_GRID]
  (void)hipGetLastError();
  for (int k0 = 0; k0 < k; k0 += DOT_VECS) {
    DotArgs args;
    std::memset(&args, 0, sizeof(args));
    args.nvec = std::min(DOT_VECS, k - k0);
    args.nblk = b->nb;
    for (int j = 0; j < b->nb; ++j) args.b[j] = b->blk[j];
    for (int v = 0; v < args.nvec; ++v) {
      if (!as[k0 + v] || as[k0 + v]->nb != b->nb || as[k0 + v]->ctx != c) return STFEM_ERR_INVALID_ARGUMENT;
      for (int j = 0; j < b->nb; ++j) args.a[v][j] = as[k0 + v]->blk[j];
    }
    if (c->prec) hipLaunchKernelGGL(multi_dot_kernel<float>, dim3(grid), dim3(256), 0, st, n_own, args, partial);
    else hipLaunchKernelGGL(multi_dot_kernel<double>, dim3(grid), dim3(256), 0, st, n_own, args, partial);
    hipLaunchKernelGGL(dot_finish_kernel, dim3(1), dim3(256), 0, st, args.nvec, grid, partial, d_out + k0);
  }
  return hipGetLastError() == hipSuccess ? STFEM_OK : STFEM_ERR_HIP;
}

} // extern "C++"

int stfem_dot(stfem_ctx *c, const stfem_vec *a, const stfem_vec *b, int64_t n_own, double *out, void *stream)
{
  if (!c || !a || !b || !out || a->nb != b->nb) return STFEM_ERR_INVALID_ARGUMENT;
  if (n_own <= 0 || n_own > c->ndofs) n_own = c->ndofs;
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (a->nb > MAX_BLOCKS) { // (vectors of more than eight blocks: eight at a time)
    double sum = 0.0;
    for (int j0 = 0; j0 < a->nb; j0 += MAX_BLOCKS) {
      stfem_vec va = *a, vb = *b;
      va.nb = vb.nb = std::min(MAX_BLOCKS, a->nb - j0);
      va.blk.assign(a->blk.begin() + j0, a->blk.begin() + j0 + va.nb);
      vb.blk.assign(b->blk.begin() + j0, b->blk.begin() + j0 + vb.nb);
      double part = 0.0;
      const int rc = stfem_dot(c, &va, &vb, n_own, &part, stream);
      if (rc != STFEM_OK) return rc;
      sum += part;
    }
    *out = sum;
    return STFEM_OK;
  }
  const stfem_vec *as[1] = {a};
  const int rc = multi_dot_device(c, 1, as, b, n_own, c->d_scratch, st);
  if (rc != STFEM_OK) return rc == STFEM_ERR_HIP ? hip_fail(hipGetLastError(), "dot") : rc;
  HIP_TRY(hipMemcpyAsync(out, c->d_scratch, sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return STFEM_OK;
}

int stfem_multi_dot(stfem_ctx *c, int k, const stfem_vec *const *as, const stfem_vec *b, int64_t n_own, double *out, void *stream)
{
  if (!c || !as || !b || !out || k < 1 || k > 256 - 8 || b->ctx != c) return STFEM_ERR_INVALID_ARGUMENT;
  if (n_own <= 0 || n_own > c->ndofs) n_own = c->ndofs;
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int rc = multi_dot_device(c, k, as, b, n_own, c->d_scratch, st);
  if (rc != STFEM_OK) return rc == STFEM_ERR_HIP ? hip_fail(hipGetLastError(), "multi_dot") : rc;
  HIP_TRY(hipMemcpyAsync(out, c->d_scratch, sizeof(double) * k, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return STFEM_OK;
}

int stfem_multi_axpy(stfem_ctx *c, int k, const double *coef, const stfem_vec *const *xs, stfem_vec *y, void *stream)
{
  if (!c || !coef || !xs || !y || k < 1 || y->ctx != c) return STFEM_ERR_INVALID_ARGUMENT;
  if (y->nb > MAX_BLOCKS) return STFEM_ERR_UNSUPPORTED;
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned grid = (unsigned)std::min<int64_t>((c->ndofs + 255) / 256, 2048);
  (void)hipGetLastError();
  for (int k0 = 0; k0 < k; k0 += DOT_VECS) {
    MultiAxpyArgs args;
    std::memset(&args, 0, sizeof(args));
    args.nvec = std::min(DOT_VECS, k - k0);
    args.sign = 1.0;
    for (int j = 0; j < y->nb; ++j) args.y[j] = y->blk[j];
    for (int v = 0; v < args.nvec; ++v) {
      if (!xs[k0 + v] || xs[k0 + v]->nb != y->nb || xs[k0 + v]->ctx != c || xs[k0 + v] == y) return STFEM_ERR_INVALID_ARGUMENT;
      args.coef[v] = coef[k0 + v];
      for (int j = 0; j < y->nb; ++j) args.x[v][j] = xs[k0 + v]->blk[j];
    }
    if (c->prec) hipLaunchKernelGGL(multi_axpy_kernel<float>, dim3(grid, y->nb), dim3(256), 0, st, c->ndofs, args);
    else hipLaunchKernelGGL(multi_axpy_kernel<double>, dim3(grid, y->nb), dim3(256), 0, st, c->ndofs, args);
  }
  return hipGetLastError() == hipSuccess ? STFEM_OK : hip_fail(hipGetLastError(), "multi_axpy");
}

// One classical Gram-Schmidt pass of w against v_0 .. v_{k-1} entirely on the device: h = V^T w (two-stage reduction),
// w -= V h with the coefficients read from device memory, h copied to the host at the end (one synchronisation).
int stfem_orthogonalize(stfem_ctx *c, int k, const stfem_vec *const *vs, stfem_vec *w, int64_t n_own, double *h_out, double *norm2_before,
                        double *norm2_out, void *stream)
{
  if (!c || !vs || !w || !h_out || k < 1 || k > 256 - 9 || w->ctx != c) return STFEM_ERR_INVALID_ARGUMENT;
  if (w->nb > MAX_BLOCKS) return STFEM_ERR_UNSUPPORTED;
  if (n_own <= 0 || n_own > c->ndofs) n_own = c->ndofs;
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  // slots of the scratch array: [0, k) coefficients, k: <w, w> before (rides in the same launch as the coefficients), k + 1: after
  std::vector<const stfem_vec *> all(vs, vs + k);
  if (norm2_before) all.push_back(w);
  int rc = multi_dot_device(c, int(all.size()), all.data(), w, n_own, c->d_scratch, st);
  if (rc != STFEM_OK) return rc == STFEM_ERR_HIP ? hip_fail(hipGetLastError(), "orthogonalize") : rc;
  const unsigned grid = (unsigned)std::min<int64_t>((c->ndofs + 255) / 256, 2048);
  for (int k0 = 0; k0 < k; k0 += DOT_VECS) {
    MultiAxpyArgs args;
    std::memset(&args, 0, sizeof(args));
    args.nvec = std::min(DOT_VECS, k - k0);
    args.sign = -1.0;
    args.dcoef = c->d_scratch + k0;
    for (int j = 0; j < w->nb; ++j) args.y[j] = w->blk[j];
    for (int v = 0; v < args.nvec; ++v) {
      if (vs[k0 + v] == w) return STFEM_ERR_ALIAS;
      for (int j = 0; j < w->nb; ++j) args.x[v][j] = vs[k0 + v]->blk[j];
    }
    if (c->prec) hipLaunchKernelGGL(multi_axpy_kernel<float>, dim3(grid, w->nb), dim3(256), 0, st, c->ndofs, args);
    else hipLaunchKernelGGL(multi_axpy_kernel<double>, dim3(grid, w->nb), dim3(256), 0, st, c->ndofs, args);
  }
  if (norm2_out) { // <w, w> after the projection, in the slot behind the coefficients
    const stfem_vec *ws[1] = {w};
    rc = multi_dot_device(c, 1, ws, w, n_own, c->d_scratch + k + 1, st);
    if (rc != STFEM_OK) return rc == STFEM_ERR_HIP ? hip_fail(hipGetLastError(), "orthogonalize") : rc;
  }
  if (hipGetLastError() != hipSuccess) return hip_fail(hipGetLastError(), "orthogonalize");
  std::vector<double> host(size_t(k) + 2);
  HIP_TRY(hipMemcpyAsync(host.data(), c->d_scratch, sizeof(double) * (k + 2), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  for (int i = 0; i < k; ++i) h_out[i] = host[i];
  if (norm2_before) *norm2_before = host[k];
  if (norm2_out) *norm2_out = host[k + 1];
  return STFEM_OK;
}

extern "C++" {
template <typename T>
__global__ __launch_bounds__(256) void plane_copy_kernel(int64_t n, const T *src, T *dst, int add)
{
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x)
    dst[i] = add ? dst[i] + src[i] : src[i];
}

} // extern "C++"

int stfem_plane_pack(stfem_ctx *c, const stfem_vec *v, int iz, void *buf, void *stream)
{
  if (!c || !v || !buf || iz < 0 || iz >= c->nd[2]) return STFEM_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  const int64_t plane = int64_t(c->nd[0]) * c->nd[1];
  for (int b = 0; b < v->nb; ++b)
    HIP_TRY(hipMemcpyAsync(static_cast<char *>(buf) + size_t(b) * plane * c->es,
                           static_cast<const char *>(v->blk[b]) + size_t(plane) * iz * c->es, plane * c->es,
                           hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
  return STFEM_OK;
}

int stfem_plane_unpack(stfem_ctx *c, stfem_vec *v, int iz, const void *buf, int add, void *stream)
{
  if (!c || !v || !buf || iz < 0 || iz >= c->nd[2]) return STFEM_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(c->device));
  const int64_t plane = int64_t(c->nd[0]) * c->nd[1];
  const unsigned grid = (unsigned)std::min<int64_t>((plane + 255) / 256, 4096);
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int b = 0; b < v->nb; ++b) {
    if (c->prec)
      hipLaunchKernelGGL(plane_copy_kernel<float>, dim3(grid), dim3(256), 0, st, plane,
                         static_cast<const float *>(buf) + b * plane, static_cast<float *>(v->blk[b]) + plane * iz, add);
    else
      hipLaunchKernelGGL(plane_copy_kernel<double>, dim3(grid), dim3(256), 0, st, plane,
                         static_cast<const double *>(buf) + b * plane, static_cast<double *>(v->blk[b]) + plane * iz, add);
  }
  return STFEM_OK;
}

extern "C++" {
struct PlanesMoveArgs {
  const void *src[MAX_BLOCKS];
  void *dst[MAX_BLOCKS];
};
// blockIdx.y = plane, blockIdx.z = block; the first / last plane of the range may be added to the destination instead of copied
template <typename T>
__global__ __launch_bounds__(256) void planes_move_kernel(int64_t plane, int nplanes, int add_mask, const PlanesMoveArgs a)
{
  const int q = blockIdx.y;
  const bool add = (q == 0 && (add_mask & 1)) || (q == nplanes - 1 && (add_mask & 2));
  const T *s = static_cast<const T *>(a.src[blockIdx.z]) + plane * q;
  T *d = static_cast<T *>(a.dst[blockIdx.z]) + plane * q;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < plane; i += int64_t(gridDim.x) * blockDim.x) d[i] = add ? d[i] + s[i] : s[i];
}
} // extern "C++"

// nplanes consecutive DoF planes of src (from plane iz_src, context cs) to dst (from plane iz_dst, context cd: the same plane size and
// Number); add_mask bit 0: the first plane is ADDED to the destination's, bit 1: the last one.  One launch for all blocks.
int stfem_planes_move(stfem_ctx *cs, const stfem_vec *src, int iz_src, stfem_ctx *cd, stfem_vec *dst, int iz_dst, int nplanes, int add_mask,
                      void *stream)
{
  if (!cs || !cd || !src || !dst || src->ctx != cs || dst->ctx != cd || nplanes < 1 || iz_src < 0 || iz_dst < 0 || iz_src + nplanes > cs->nd[2] ||
      iz_dst + nplanes > cd->nd[2])
    return STFEM_ERR_INVALID_ARGUMENT;
  if (cs->nd[0] != cd->nd[0] || cs->nd[1] != cd->nd[1] || cs->prec != cd->prec || cs->device != cd->device || src->nb != dst->nb)
    return STFEM_ERR_SHAPE_MISMATCH;
  if (src->nb > MAX_BLOCKS) return STFEM_ERR_UNSUPPORTED;
  HIP_TRY(hipSetDevice(cd->device));
  const int64_t plane = int64_t(cd->nd[0]) * cd->nd[1];
  PlanesMoveArgs a;
  std::memset(&a, 0, sizeof(a));
  for (int b = 0; b < src->nb; ++b) {
    a.src[b] = static_cast<const char *>(src->blk[b]) + size_t(plane) * iz_src * cs->es;
    a.dst[b] = static_cast<char *>(dst->blk[b]) + size_t(plane) * iz_dst * cd->es;
  }
  const dim3 grid((unsigned)std::min<int64_t>((plane + 255) / 256, 1024), nplanes, src->nb);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (cd->prec) hipLaunchKernelGGL(planes_move_kernel<float>, grid, dim3(256), 0, st, plane, nplanes, add_mask, a);
  else hipLaunchKernelGGL(planes_move_kernel<double>, grid, dim3(256), 0, st, plane, nplanes, add_mask, a);
  if (hipGetLastError() != hipSuccess) return hip_fail(hipGetLastError(), "planes_move_kernel");
  return STFEM_OK;
}

// ------------------------------------------------------------------------------------ host helpers

// time-multigrid transfer matrices (fe_time.h:749-898); out may be NULL to ask for the dimensions only
static int time_transfer_out(int rc, const Mat &M, int m, int n, double *out, int32_t dims[2])
{
  if (rc != 0 || !dims) return STFEM_ERR_INVALID_ARGUMENT;
  dims[0] = m;
  dims[1] = n;
  if (out) std::copy(M.begin(), M.end(), out);
  return STFEM_OK;
}
int stfem_time_prolongation_matrix(int type, int r, int n_timesteps_at_once, double *out, int32_t dims[2])
{
  Mat M;
  int m = 0, n = 0;
  const int rc = time_prolongation(type, r, n_timesteps_at_once, M, m, n);
  return time_transfer_out(rc, M, m, n, out, dims);
}
int stfem_time_restriction_matrix(int type, int r, int n_timesteps_at_once, double *out, int32_t dims[2])
{
  Mat M;
  int m = 0, n = 0;
  const int rc = time_restriction(type, r, n_timesteps_at_once, M, m, n);
  return time_transfer_out(rc, M, m, n, out, dims);
}
int stfem_time_projection_matrix(int type, int r_src, int r_dst, int n_timesteps_at_once, double *out, int32_t dims[2])
{
  Mat M;
  int m = 0, n = 0;
  const int rc = time_projection(type, r_src, r_dst, n_timesteps_at_once, M, m, n);
  return time_transfer_out(rc, M, m, n, out, dims);
}

int stfem_fe_time_weights(int type, int r, double tau, int ns, double *Alpha, double *Beta,
                          double *Gamma, double *Zeta)
{
  if ((type != 0 && type != 1) || ns < 1 || !Alpha || !Beta || !Gamma || !Zeta || r > 8)
    return STFEM_ERR_INVALID_ARGUMENT;
  try {
    Mat A, B, G, Z;
    const int nb = fe_time_weights(type, r, tau, ns, A, B, G, Z);
    std::copy(A.begin(), A.end(), Alpha);
    std::copy(B.begin(), B.end(), Beta);
    std::copy(G.begin(), G.end(), Gamma);
    std::copy(Z.begin(), Z.end(), Zeta);
    return nb;
  } catch (...) {
    return STFEM_ERR_INVALID_ARGUMENT;
  }
}

int stfem_fe_time_weights_wave(int type, int r, double tau, int ns, double *AL, double *BL,
                               double *uK, double *uM, double *vM)
{
  if ((type != 0 && type != 1) || ns < 1 || !AL || !BL || !uK || !uM || !vM || r > 8)
    return STFEM_ERR_INVALID_ARGUMENT;
  try {
    Mat a, b, k, m, v;
    const int nb = fe_time_weights_wave(type, r, tau, ns, a, b, k, m, v);
    std::copy(a.begin(), a.end(), AL);
    std::copy(b.begin(), b.end(), BL);
    std::copy(k.begin(), k.end(), uK);
    std::copy(m.begin(), m.end(), uM);
    std::copy(v.begin(), v.end(), vM);
    return nb;
  } catch (...) {
    return STFEM_ERR_INVALID_ARGUMENT;
  }
}

int stfem_mesh_vertices(const int32_t gn[3], const double lo[3], const double up[3], double distort,
                        uint64_t seed, int32_t z0, int32_t z1, double *out)
{
  if (!gn || !lo || !up || !out || z0 < 0 || z1 > gn[2] || z0 >= z1) return STFEM_ERR_INVALID_ARGUMENT;
  mesh_vertices(gn, lo, up, distort, seed, z0, z1, out);
  return STFEM_OK;
}

int stfem_coefficient_per_cell(const int32_t nc[3], const double *vertices, double c1, double c2,
                               double c3, double distort, const int32_t sub[3], const double lo[3],
                               const double up[3], double *out)
{
  if (!nc || !vertices || !sub || !lo || !up || !out) return STFEM_ERR_INVALID_ARGUMENT;
  coefficient_per_cell(nc, vertices, c1, c2, c3, distort, sub, lo, up, out);
  return STFEM_OK;
}

} // extern "C"
