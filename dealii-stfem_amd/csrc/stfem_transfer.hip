// Space transfers of the space-time multigrid: what the reference gets from deal.II's
// MGTwoLevelTransfer<dim, VectorT<Number>> (reinit / prolongate_and_add / restrict_and_add / interpolate as used by
// MGTwoLevelBlockTransfer, include/stmg.h:38-110, built in build_stmg_transfers, stmg.h:580-600) for the
// structured meshes of this library: h-transfer (twice the cells per direction, same degree) and p-transfer (same
// cells, lower degree), with the zero-boundary constraints of both levels.
//
// MI355X design: on a structured block with lexicographic numbering the embedding of the coarse space is the
// Kronecker product Pz (x) Py (x) Px of banded 1D matrices (the transfer is defined on the reference cell, so this
// holds for MappingQ1-perturbed meshes too; a DoF is constrained iff one of its three line indices is, so the
// constraints factorise as well).  A transfer is therefore three passes of one kernel, a banded 1D mat-vec along
// one axis of the array with the x index on the lanes: every load and store is a coalesced full line, there is no
// scatter, no atomics and no colouring, and the result is reproducible.  HBM traffic of an h-prolongation:
// 29 coarse-vector sizes against the 17 a single fused pass would need (read coarse, read + write fine).
// Also here: the level schedule of the multigrid (fe_time.cc:40-150) and the precision change between the solver's
// vectors and the multigrid's (GMG::vmult, stmg.h:1330-1343: copy_locally_owned_data_from).
#include "stfem_internal.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {

thread_local char g_transfer_err[256] = "";
#define TR_TRY(call)                                                                            \
  do {                                                                                          \
    hipError_t e_ = (call);                                                                     \
    if (e_ != hipSuccess) {                                                                     \
      snprintf(g_transfer_err, sizeof(g_transfer_err), "%s: %s", #call, hipGetErrorString(e_)); \
      return STFEM_ERR_HIP;                                                                     \
    }                                                                                           \
  } while (0)

// one banded 1D matrix in row-compressed form: row o = sum_k w[off[o] + k] * in[first[o] + k], k < off[o+1] - off[o]
struct Band {
  int n_out = 0, n_in = 0;
  std::vector<int> first, off;
  std::vector<double> w;
  int *d_first = nullptr, *d_off = nullptr;
  double *d_w = nullptr;
};

// dense n_out x n_in -> Band (the nonzeros of a row are contiguous up to exact zeros in between)
Band make_band(int n_out, int n_in, const std::vector<double> &A)
{
  Band b;
  b.n_out = n_out;
  b.n_in = n_in;
  b.off.push_back(0);
  for (int o = 0; o < n_out; ++o) {
    int lo = n_in, hi = -1;
    for (int i = 0; i < n_in; ++i)
      if (A[size_t(o) * n_in + i] != 0.0) {
        lo = std::min(lo, i);
        hi = std::max(hi, i);
      }
    b.first.push_back(hi < 0 ? 0 : lo);
    for (int i = lo; i <= hi; ++i) b.w.push_back(A[size_t(o) * n_in + i]);
    b.off.push_back(int(b.w.size()));
  }
  if (b.w.empty()) b.w.push_back(0.0);
  return b;
}

// line index -> (cell, local node) of a 1D FE_Q(p) line with n cells
void cell_of(int i, int p, int n, int &cell, int &j)
{
  if (i == p * n) {
    cell = n - 1;
    j = p;
  } else {
    cell = i / p;
    j = i % p;
  }
}

// 1D embedding P[n_f x n_c] (coarse nodal values -> fine nodal values of the same function) and the nodal
// interpolation I[n_c x n_f] (the fine function at the coarse nodes); r = fine cells per coarse cell
void line_matrices(int nc_f, int p_f, int nc_c, int p_c, std::vector<double> &P, std::vector<double> &I)
{
  const int r = nc_f / nc_c, n_f = p_f * nc_f + 1, n_c = p_c * nc_c + 1;
  const std::vector<double> gf = stfem::lobatto_points(p_f + 1), gc = stfem::lobatto_points(p_c + 1);
  P.assign(size_t(n_f) * n_c, 0.0);
  I.assign(size_t(n_c) * n_f, 0.0);
  stfem::Mat V, G;
  for (int f = 0; f < n_f; ++f) {
    int cell, j;
    cell_of(f, p_f, nc_f, cell, j);
    const int ec = cell / r;
    const double xi = (double(cell % r) + gf[j]) / r;
    stfem::lagrange_tables(gc, {xi}, V, G);
    for (int a = 0; a <= p_c; ++a) P[size_t(f) * n_c + ec * p_c + a] = std::abs(V[a]) < 1e-15 ? 0.0 : V[a];
  }
  for (int c = 0; c < n_c; ++c) {
    int cell, j;
    cell_of(c, p_c, nc_c, cell, j);
    const double t = gc[j] * r;
    const int sub = std::min(int(t), r - 1);
    stfem::lagrange_tables(gf, {t - sub}, V, G);
    for (int a = 0; a <= p_f; ++a) I[size_t(c) * n_f + (cell * r + sub) * p_f + a] = std::abs(V[a]) < 1e-15 ? 0.0 : V[a];
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
axis_apply_kernel(T *__restrict__ out, const T *__restrict__ in, int d0, int d1, int d2, int axis, int n_in, const int *__restrict__ first,
                  const int *__restrict__ off, const double *__restrict__ w, int add)
{
  const long long total = (long long)d0 * d1 * d2;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int i0 = int(t % d0), i1 = int((t / d0) % d1), i2 = int(t / ((long long)d0 * d1));
    const int o = axis == 0 ? i0 : axis == 1 ? i1 : i2;
    const int k0 = off[o], k1 = off[o + 1], f = first[o];
    long long base, stride;
    if (axis == 0) {
      base = f + (long long)n_in * (i1 + (long long)d1 * i2);
      stride = 1;
    } else if (axis == 1) {
      base = i0 + (long long)d0 * (f + (long long)n_in * i2);
      stride = d0;
    } else {
      base = i0 + (long long)d0 * (i1 + (long long)d1 * f);
      stride = (long long)d0 * d1;
    }
    T acc = add ? out[t] : T(0);
    for (int k = k0; k < k1; ++k) acc += T(w[k]) * in[base + (k - k0) * stride];
    out[t] = acc;
  }
}

template <typename TD, typename TS> __global__ void convert_kernel(TD *__restrict__ d, const TS *__restrict__ s, long long n)
{
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) d[t] = TD(s[t]);
}

} // namespace

struct stfem_transfer {
  stfem_ctx *fine = nullptr, *coarse = nullptr;
  Band P[3], R[3], I[3]; // per direction: prolongation rows, its transpose, nodal interpolation (all with the constraints)
  void *d_tmp[2] = {nullptr, nullptr};
  size_t tmp_elems = 0;
};

namespace {

int upload(Band &b)
{
  TR_TRY(hipMalloc(&b.d_first, b.first.size() * sizeof(int)));
  TR_TRY(hipMalloc(&b.d_off, b.off.size() * sizeof(int)));
  TR_TRY(hipMalloc(&b.d_w, b.w.size() * sizeof(double)));
  TR_TRY(hipMemcpy(b.d_first, b.first.data(), b.first.size() * sizeof(int), hipMemcpyHostToDevice));
  TR_TRY(hipMemcpy(b.d_off, b.off.data(), b.off.size() * sizeof(int), hipMemcpyHostToDevice));
  TR_TRY(hipMemcpy(b.d_w, b.w.data(), b.w.size() * sizeof(double), hipMemcpyHostToDevice));
  return STFEM_OK;
}
void release(Band &b)
{
  (void)hipFree(b.d_first);
  (void)hipFree(b.d_off);
  (void)hipFree(b.d_w);
}

template <typename T>
int launch_axis(T *out, const T *in, const int dims[3], int axis, const Band &b, int add, hipStream_t s)
{
  const long long total = (long long)dims[0] * dims[1] * dims[2];
  const int blocks = int(std::min<long long>((total + 255) / 256, 1 << 20));
  axis_apply_kernel<T><<<blocks, 256, 0, s>>>(out, in, dims[0], dims[1], dims[2], axis, b.n_in, b.d_first, b.d_off, b.d_w, add);
  TR_TRY(hipGetLastError());
  return STFEM_OK;
}

// out (dims of `to`) (+)= (B2 (x) B1 (x) B0) in; order: the axes in `order`, smallest intermediates first
template <typename T>
int apply3(stfem_transfer *t, const Band B[3], void *out, const void *in, const int order[3], int add, hipStream_t s)
{
  int dims[3] = {B[0].n_in, B[1].n_in, B[2].n_in};
  const T *cur = static_cast<const T *>(in);
  for (int step = 0; step < 3; ++step) {
    const int ax = order[step];
    dims[ax] = B[ax].n_out;
    T *dst = step == 2 ? static_cast<T *>(out) : static_cast<T *>(t->d_tmp[step]);
    const int st = launch_axis<T>(dst, cur, dims, ax, B[ax], step == 2 ? add : 0, s);
    if (st != STFEM_OK) return st;
    cur = dst;
  }
  return STFEM_OK;
}

int run(stfem_transfer *t, const Band B[3], stfem_ctx *to, stfem_ctx *from, stfem_vec *dst, const stfem_vec *src, bool expanding, int add,
        void *stream)
{
  if (!t || !dst || !src) return STFEM_ERR_INVALID_ARGUMENT;
  if (dst->ctx != to || src->ctx != from || dst->nb != src->nb) return STFEM_ERR_SHAPE_MISMATCH;
  // expanding (prolongation): x, y, z keeps the intermediates small; contracting: z, y, x
  const int up[3] = {0, 1, 2}, down[3] = {2, 1, 0};
  hipStream_t s = static_cast<hipStream_t>(stream);
  for (int b = 0; b < dst->nb; ++b) {
    const int st = to->prec == 0 ? apply3<double>(t, B, dst->blk[b], src->blk[b], expanding ? up : down, add, s)
                                 : apply3<float>(t, B, dst->blk[b], src->blk[b], expanding ? up : down, add, s);
    if (st != STFEM_OK) return st;
  }
  return STFEM_OK;
}

} // namespace

extern "C" {

const char *stfem_transfer_last_error(void) { return g_transfer_err; }

int stfem_transfer_create(stfem_ctx *fine, stfem_ctx *coarse, stfem_transfer **out)
{
  if (!fine || !coarse || !out) return STFEM_ERR_INVALID_ARGUMENT;
  if (fine->prec != coarse->prec || fine->device != coarse->device) return STFEM_ERR_SHAPE_MISMATCH;
  for (int d = 0; d < 3; ++d) {
    const bool same = fine->nc[d] == coarse->nc[d], twice = fine->nc[d] == 2 * coarse->nc[d];
    if (!same && !twice) return STFEM_ERR_SHAPE_MISMATCH;
  }
  if (coarse->p > fine->p) return STFEM_ERR_SHAPE_MISMATCH;
  TR_TRY(hipSetDevice(fine->device));
  stfem_transfer *t = new stfem_transfer;
  t->fine = fine;
  t->coarse = coarse;
  for (int d = 0; d < 3; ++d) {
    std::vector<double> P, I;
    line_matrices(fine->nc[d], fine->p, coarse->nc[d], coarse->p, P, I);
    const int n_f = fine->nd[d], n_c = coarse->nd[d];
    // zero-boundary constraints of both levels: constrained rows are not written, constrained columns read as 0
    auto constrained = [&](const stfem_ctx *c, int i, int n) { return (i == 0 && (c->dmask >> (2 * d) & 1)) || (i == n - 1 && (c->dmask >> (2 * d + 1) & 1)); };
    std::vector<double> R(size_t(n_c) * n_f);
    for (int f = 0; f < n_f; ++f)
      for (int c = 0; c < n_c; ++c) {
        if (constrained(fine, f, n_f) || constrained(coarse, c, n_c)) P[size_t(f) * n_c + c] = 0.0, I[size_t(c) * n_f + f] = 0.0;
        R[size_t(c) * n_f + f] = P[size_t(f) * n_c + c];
      }
    t->P[d] = make_band(n_f, n_c, P);
    t->R[d] = make_band(n_c, n_f, R);
    t->I[d] = make_band(n_c, n_f, I);
    for (Band *b : {&t->P[d], &t->R[d], &t->I[d]}) {
      const int st = upload(*b);
      if (st != STFEM_OK) {
        stfem_transfer_destroy(t);
        return st;
      }
    }
  }
  // intermediates: (fine x, coarse y, coarse z) and (fine x, fine y, coarse z); the contracting order needs
  // (fine x, fine y, coarse z) and (fine x, coarse y, coarse z): the larger of the two fits both roles
  t->tmp_elems = size_t(fine->nd[0]) * fine->nd[1] * coarse->nd[2];
  for (int k = 0; k < 2; ++k)
    if (hipMalloc(&t->d_tmp[k], t->tmp_elems * fine->es) != hipSuccess) {
      stfem_transfer_destroy(t);
      return STFEM_ERR_OUT_OF_MEMORY;
    }
  *out = t;
  return STFEM_OK;
}

void stfem_transfer_destroy(stfem_transfer *t)
{
  if (!t) return;
  for (int d = 0; d < 3; ++d) {
    release(t->P[d]);
    release(t->R[d]);
    release(t->I[d]);
  }
  (void)hipFree(t->d_tmp[0]);
  (void)hipFree(t->d_tmp[1]);
  delete t;
}

int stfem_transfer_prolongate(stfem_transfer *t, stfem_vec *dst_fine, const stfem_vec *src_coarse, int add, void *stream)
{
  return t ? run(t, t->P, t->fine, t->coarse, dst_fine, src_coarse, true, add, stream) : STFEM_ERR_INVALID_ARGUMENT;
}
int stfem_transfer_restrict(stfem_transfer *t, stfem_vec *dst_coarse, const stfem_vec *src_fine, int add, void *stream)
{
  return t ? run(t, t->R, t->coarse, t->fine, dst_coarse, src_fine, false, add, stream) : STFEM_ERR_INVALID_ARGUMENT;
}
int stfem_transfer_interpolate(stfem_transfer *t, stfem_vec *dst_coarse, const stfem_vec *src_fine, void *stream)
{
  return t ? run(t, t->I, t->coarse, t->fine, dst_coarse, src_fine, false, 0, stream) : STFEM_ERR_INVALID_ARGUMENT;
}

int stfem_transfer_line_matrices(int ncell_fine, int degree_fine, int ncell_coarse, int degree_coarse, double *P, double *I)
{
  if (ncell_coarse < 1 || degree_coarse < 1 || degree_fine < degree_coarse || (ncell_fine != ncell_coarse && ncell_fine != 2 * ncell_coarse))
    return STFEM_ERR_INVALID_ARGUMENT;
  std::vector<double> p, i;
  line_matrices(ncell_fine, degree_fine, ncell_coarse, degree_coarse, p, i);
  if (P) std::copy(p.begin(), p.end(), P);
  if (I) std::copy(i.begin(), i.end(), I);
  return STFEM_OK;
}

int stfem_vector_convert(stfem_vec *dst, const stfem_vec *src, void *stream)
{
  if (!dst || !src) return STFEM_ERR_INVALID_ARGUMENT;
  if (dst->nb != src->nb || dst->ctx->ndofs != src->ctx->ndofs) return STFEM_ERR_SHAPE_MISMATCH;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const long long n = dst->ctx->ndofs;
  const int blocks = int(std::min<long long>((n + 255) / 256, 1 << 16));
  for (int b = 0; b < dst->nb; ++b) {
    const int pd = dst->ctx->prec, ps = src->ctx->prec;
    if (pd == ps) TR_TRY(hipMemcpyAsync(dst->blk[b], src->blk[b], size_t(n) * dst->ctx->es, hipMemcpyDeviceToDevice, s));
    else if (pd == 1) convert_kernel<float, double><<<blocks, 256, 0, s>>>(static_cast<float *>(dst->blk[b]), static_cast<const double *>(src->blk[b]), n);
    else convert_kernel<double, float><<<blocks, 256, 0, s>>>(static_cast<double *>(dst->blk[b]), static_cast<const float *>(src->blk[b]), n);
  }
  TR_TRY(hipGetLastError());
  return STFEM_OK;
}

int stfem_poly_mg_sequence(int k_max, int k_min, int sequence_type, int32_t *out, int32_t *n_out)
{
  if (!n_out || k_min < 0 || k_max < k_min) return STFEM_ERR_INVALID_ARGUMENT;
  const std::vector<int> s = stfem::poly_mg_sequence(k_max, k_min, sequence_type);
  if (s.empty()) return STFEM_ERR_INVALID_ARGUMENT;
  if (out) std::copy(s.begin(), s.end(), out);
  *n_out = int32_t(s.size());
  return STFEM_OK;
}

int stfem_mg_sequence(int n_sp_lvl, int n_k, int n_p, int n_timesteps_at_once, int n_timesteps_at_once_min, char lower_lvl, int coarsening_type,
                      int time_before_space, int use_p_multigrid_space, int zip_from_back, char *out, int32_t *n_out)
{
  if (!n_out || n_sp_lvl < 1 || n_k < 1 || (use_p_multigrid_space && n_p < 1) || n_timesteps_at_once < 1 || n_timesteps_at_once_min < 1 ||
      (lower_lvl != 'k' && lower_lvl != 't'))
    return STFEM_ERR_INVALID_ARGUMENT;
  const std::string s = stfem::mg_sequence(n_sp_lvl, n_k, n_p, n_timesteps_at_once, n_timesteps_at_once_min, lower_lvl, coarsening_type,
                                           time_before_space != 0, use_p_multigrid_space != 0, zip_from_back != 0);
  if (out) std::copy(s.begin(), s.end(), out);
  *n_out = int32_t(s.size());
  return STFEM_OK;
}

int stfem_precondition_stmg_types(const char *mg_type_level, int n, int coarsening_type, int time_before_space, int smoother, int32_t *out)
{
  if (!mg_type_level || !out || n < 0) return STFEM_ERR_INVALID_ARGUMENT;
  const std::vector<int> r = stfem::precondition_stmg_types(std::string(mg_type_level, size_t(n)), coarsening_type, time_before_space != 0, smoother);
  std::copy(r.begin(), r.end(), out);
  return STFEM_OK;
}

} // extern "C"
