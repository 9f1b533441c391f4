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
#include <cstdlib>
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


// Cell form of the 1D passes along y and z (the bulk of the traffic): a thread takes one COARSE cell of a line - the local
// embedding matrix L [(R + 1) x (PC + 1)] is the same for every cell, so the weights are kernel arguments (scalar registers),
// the PC + 1 coarse values are loaded once for the R (+ 1) fine values they produce, and no index table is read.
// Addressing: element (inner, i, outer) of an array with n entries along the axis at inner + S (i + n outer), inner < S contiguous.
template <typename T> struct CellMat {
  T L[9 * 5]; // L[j * (PC + 1) + a]
};
constexpr int CF_LO_C = 1, CF_HI_C = 2, CF_LO_F = 4, CF_HI_F = 8; // constrained ends of the coarse / fine line

template <typename T, int PC, int R>
__global__ void __launch_bounds__(256)
cell_prolongate_kernel(T *__restrict__ out, const T *__restrict__ in, long long S, int ncell, long long total, const CellMat<T> m, int flags, int add)
{
  const long long n_c = (long long)PC * ncell + 1, n_f = (long long)R * ncell + 1;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const long long inner = t % S, rest = t / S;
    const int c = int(rest % ncell);
    const long long outer = rest / ncell;
    const bool first = c == 0, last = c == ncell - 1;
    T u[PC + 1];
#pragma unroll
    for (int a = 0; a <= PC; ++a) u[a] = in[inner + S * (c * (long long)PC + a + n_c * outer)];
    if (first && (flags & CF_LO_C)) u[0] = T(0);
    if (last && (flags & CF_HI_C)) u[PC] = T(0);
    T *o = out + inner + S * (c * (long long)R + n_f * outer);
#pragma unroll
    for (int j = 0; j <= R; ++j) {
      if (j == R && !last) break; // the upper end belongs to the next cell
      T v = T(0);
#pragma unroll
      for (int a = 0; a <= PC; ++a) v += m.L[j * (PC + 1) + a] * u[a];
      const bool constrained = (first && j == 0 && (flags & CF_LO_F)) || (last && j == R && (flags & CF_HI_F));
      if (add) {
        if (!constrained) o[S * j] += v;
      } else o[S * j] = constrained ? T(0) : v;
    }
  }
}

// The passes along y and z of the prolongation in ONE kernel (same embedding matrix along both axes): a thread takes one coarse cell of
// the (y, z) plane at its x index, loads its (PC + 1)^2 coarse values once and writes the R^2 (+ the upper ends at the mesh boundary)
// fine values they produce - the (fine x, fine y, coarse z) intermediate of the three-pass form (4 coarse-vector sizes written and
// read again for an h-transfer) never exists: 21 instead of 29 coarse-vector sizes of traffic.
// in: [nx][PC ncy + 1][PC ncz + 1], out: [nx][R ncy + 1][R ncz + 1], x contiguous.
template <typename T, int PC, int R>
__global__ void __launch_bounds__(256)
cell_prolongate_yz_kernel(T *__restrict__ out, const T *__restrict__ in, int nx, int ncy, int ncz, long long total, const CellMat<T> m, int flags_y,
                          int flags_z, int add)
{
  const long long n_cy = (long long)PC * ncy + 1, n_fy = (long long)R * ncy + 1;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int x = int(t % nx);
    const long long rest = t / nx;
    const int cy = int(rest % ncy), cz = int(rest / ncy);
    const bool first_y = cy == 0, last_y = cy == ncy - 1, first_z = cz == 0, last_z = cz == ncz - 1;
    T u[PC + 1][PC + 1]; // [az][ay]
#pragma unroll
    for (int az = 0; az <= PC; ++az)
#pragma unroll
      for (int ay = 0; ay <= PC; ++ay) u[az][ay] = in[x + (long long)nx * ((cy * (long long)PC + ay) + n_cy * (cz * (long long)PC + az))];
    if (first_y && (flags_y & CF_LO_C)) {
#pragma unroll
      for (int az = 0; az <= PC; ++az) u[az][0] = T(0);
    }
    if (last_y && (flags_y & CF_HI_C)) {
#pragma unroll
      for (int az = 0; az <= PC; ++az) u[az][PC] = T(0);
    }
    if (first_z && (flags_z & CF_LO_C)) {
#pragma unroll
      for (int ay = 0; ay <= PC; ++ay) u[0][ay] = T(0);
    }
    if (last_z && (flags_z & CF_HI_C)) {
#pragma unroll
      for (int ay = 0; ay <= PC; ++ay) u[PC][ay] = T(0);
    }
    T *o = out + x + (long long)nx * (cy * (long long)R + n_fy * (cz * (long long)R));
#pragma unroll
    for (int jz = 0; jz <= R; ++jz) {
      if (jz == R && !last_z) break; // the upper ends belong to the next cells
      T tz[PC + 1];
#pragma unroll
      for (int ay = 0; ay <= PC; ++ay) {
        T v = T(0);
#pragma unroll
        for (int az = 0; az <= PC; ++az) v += m.L[jz * (PC + 1) + az] * u[az][ay];
        tz[ay] = v;
      }
      const bool con_z = (first_z && jz == 0 && (flags_z & CF_LO_F)) || (last_z && jz == R && (flags_z & CF_HI_F));
#pragma unroll
      for (int jy = 0; jy <= R; ++jy) {
        if (jy == R && !last_y) break;
        T v = T(0);
#pragma unroll
        for (int ay = 0; ay <= PC; ++ay) v += m.L[jy * (PC + 1) + ay] * tz[ay];
        const bool constrained = con_z || (first_y && jy == 0 && (flags_y & CF_LO_F)) || (last_y && jy == R && (flags_y & CF_HI_F));
        T *q = o + (long long)nx * (jy + n_fy * jz);
        if (add) {
          if (!constrained) *q += v;
        } else *q = constrained ? T(0) : v;
      }
    }
  }
}

// the transpose: coarse node a of cell c collects the fine values of its own cell and, for a = 0, of the interior of the cell before
template <typename T, int PC, int R>
__global__ void __launch_bounds__(256)
cell_restrict_kernel(T *__restrict__ out, const T *__restrict__ in, long long S, int ncell, long long total, const CellMat<T> m, int flags, int add)
{
  const long long n_c = (long long)PC * ncell + 1, n_f = (long long)R * ncell + 1;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const long long inner = t % S, rest = t / S;
    const int c = int(rest % ncell);
    const long long outer = rest / ncell;
    const bool first = c == 0, last = c == ncell - 1;
    const T *f = in + inner + S * (c * (long long)R + n_f * outer);
    T u[R + 1], w[R > 1 ? R - 1 : 1];
#pragma unroll
    for (int j = 0; j <= R; ++j) u[j] = f[S * j];
#pragma unroll
    for (int j = 1; j < R; ++j) w[j - 1] = first ? T(0) : f[S * (j - R)];
    if (first && (flags & CF_LO_F)) u[0] = T(0);
    if (last && (flags & CF_HI_F)) u[R] = T(0);
    T *o = out + inner + S * (c * (long long)PC + n_c * outer);
#pragma unroll
    for (int a = 0; a <= PC; ++a) {
      if (a == PC && !last) break;
      T v = T(0);
#pragma unroll
      for (int j = (a == PC ? 1 : 0); j <= (a == 0 ? R - 1 : R); ++j) v += m.L[j * (PC + 1) + a] * u[j];
      if (a == 0) {
#pragma unroll
        for (int j = 1; j < R; ++j) v += m.L[j * (PC + 1) + PC] * w[j - 1];
      }
      const bool constrained = (first && a == 0 && (flags & CF_LO_C)) || (last && a == PC && (flags & CF_HI_C));
      if (add) {
        if (!constrained) o[S * a] += v;
      } else o[S * a] = constrained ? T(0) : v;
    }
  }
}

// The same restriction as a march: a thread walks along the axis through the coarse cells [c0, c1) of its segment and keeps
// the interior fine values of the cell before in registers, so every fine value is loaded once (cell_restrict_kernel loads the
// previous cell's rows again: the planes of one coarse cell exceed the L2 along z, 2 x the HBM reads).
template <typename T, int PC, int R>
__global__ void __launch_bounds__(256)
cell_restrict_march_kernel(T *__restrict__ out, const T *__restrict__ in, long long S, int ncell, int nseg, long long total, const CellMat<T> m, int flags,
                           int add)
{
  const long long n_c = (long long)PC * ncell + 1, n_f = (long long)R * ncell + 1;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const long long inner = t % S, rest = t / S;
    const int seg = int(rest % nseg);
    const long long outer = rest / nseg;
    const int c0 = int((long long)ncell * seg / nseg), c1 = int((long long)ncell * (seg + 1) / nseg);
    const T *f = in + inner + S * (n_f * outer);
    T *o = out + inner + S * (n_c * outer);
    T w[R > 1 ? R - 1 : 1], u0;
#pragma unroll
    for (int j = 1; j < R; ++j) w[j - 1] = c0 > 0 ? f[S * ((long long)(c0 - 1) * R + j)] : T(0);
    u0 = f[S * ((long long)c0 * R)];
    if (c0 == 0 && (flags & CF_LO_F)) u0 = T(0);
    for (int c = c0; c < c1; ++c) {
      const bool first = c == 0, last = c == ncell - 1;
      T u[R + 1];
      u[0] = u0;
#pragma unroll
      for (int j = 1; j <= R; ++j) u[j] = f[S * ((long long)c * R + j)];
      if (last && (flags & CF_HI_F)) u[R] = T(0);
#pragma unroll
      for (int a = 0; a <= PC; ++a) {
        if (a == PC && !last) break;
        T v = T(0);
#pragma unroll
        for (int j = (a == PC ? 1 : 0); j <= (a == 0 ? R - 1 : R); ++j) v += m.L[j * (PC + 1) + a] * u[j];
        if (a == 0) {
#pragma unroll
          for (int j = 1; j < R; ++j) v += m.L[j * (PC + 1) + PC] * w[j - 1];
        }
        const bool constrained = (first && a == 0 && (flags & CF_LO_C)) || (last && a == PC && (flags & CF_HI_C));
        T *q = o + S * ((long long)c * PC + a);
        if (add) {
          if (!constrained) *q += v;
        } else *q = constrained ? T(0) : v;
      }
#pragma unroll
      for (int j = 1; j < R; ++j) w[j - 1] = u[j];
      u0 = u[R];
    }
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
  // cell form of P / R along y and z: local embedding matrix, coarse degree, fine nodes per coarse cell, coarse cells, constrained ends
  // (up to FE_Q(5) on both levels with two fine cells per coarse one: 11 x 6 entries; the cell kernels are instantiated up to
  // FE_Q(4): 9 x 5, larger blocks take the table-driven passes)
  double L[3][11 * 6] = {};
  bool cell_restrict_z = true;
  bool cell_form = true; // STFEM_TRANSFER_TABLES=1: table-driven passes along every axis (for comparison)
  int pc[3] = {0, 0, 0}, Rn[3] = {0, 0, 0}, ncc[3] = {0, 0, 0}, flags[3] = {0, 0, 0};
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


// restriction as a march along the axis with at most this many segments per line (0: one thread per coarse cell); STFEM_TRANSFER_MARCH
static int march_segments = [] {
  const char *e = getenv("STFEM_TRANSFER_MARCH");
  return e ? atoi(e) : 64;
}();

template <typename T, int PC, int R>
int launch_cell_t(bool prolongate, T *out, const T *in, long long S, int ncell, long long total, const double *L, int flags, int add, hipStream_t s)
{
  CellMat<T> m;
  for (int i = 0; i < 9 * 5; ++i) m.L[i] = T(L[i]);
  const int blocks = int(std::min<long long>((total + 255) / 256, 1 << 20));
  if (prolongate) cell_prolongate_kernel<T, PC, R><<<blocks, 256, 0, s>>>(out, in, S, ncell, total, m, flags, add);
  else if (march_segments > 0) {
    // enough threads to fill the chip: lines x segments >= ~2^18
    const long long lines = total / ncell;
    int nseg = int(std::min<long long>(ncell, std::max<long long>(1, (262144 + lines - 1) / lines)));
    nseg = std::min(nseg, march_segments);
    const long long tot = lines * nseg;
    const int bl = int(std::min<long long>((tot + 255) / 256, 1 << 20));
    cell_restrict_march_kernel<T, PC, R><<<bl, 256, 0, s>>>(out, in, S, ncell, nseg, tot, m, flags, add);
  } else cell_restrict_kernel<T, PC, R><<<blocks, 256, 0, s>>>(out, in, S, ncell, total, m, flags, add);
  TR_TRY(hipGetLastError());
  return STFEM_OK;
}

// returns 1 if there is no instantiation for (pc, R): the caller falls back to the table-driven kernel
template <typename T>
int launch_cell(bool prolongate, int pc, int R, T *out, const T *in, long long S, int ncell, long long total, const double *L, int flags, int add,
                hipStream_t s)
{
#define STFEM_CELL_CASE(PC_, R_) \
  if (pc == PC_ && R == R_) return launch_cell_t<T, PC_, R_>(prolongate, out, in, S, ncell, total, L, flags, add, s);
  STFEM_CELL_CASE(1, 2) STFEM_CELL_CASE(1, 3) STFEM_CELL_CASE(1, 4) STFEM_CELL_CASE(1, 6) STFEM_CELL_CASE(1, 8)
  STFEM_CELL_CASE(2, 3) STFEM_CELL_CASE(2, 4) STFEM_CELL_CASE(2, 6) STFEM_CELL_CASE(2, 8)
  STFEM_CELL_CASE(3, 4) STFEM_CELL_CASE(3, 6) STFEM_CELL_CASE(3, 8)
  STFEM_CELL_CASE(4, 8)
#undef STFEM_CELL_CASE
  return 1;
}

template <typename T>
int launch_cell_yz(int pc, int R, T *out, const T *in, int nx, int ncy, int ncz, const double *L, int flags_y, int flags_z, int add, hipStream_t s)
{
  CellMat<T> m;
  for (int i = 0; i < 9 * 5; ++i) m.L[i] = T(L[i]);
  const long long total = (long long)nx * ncy * ncz;
  const int blocks = int(std::min<long long>((total + 255) / 256, 1 << 20));
#define STFEM_CELL_CASE(PC_, R_)                                                                                              \
  if (pc == PC_ && R == R_) {                                                                                                 \
    cell_prolongate_yz_kernel<T, PC_, R_><<<blocks, 256, 0, s>>>(out, in, nx, ncy, ncz, total, m, flags_y, flags_z, add);     \
    TR_TRY(hipGetLastError());                                                                                                \
    return STFEM_OK;                                                                                                          \
  }
  STFEM_CELL_CASE(1, 2) STFEM_CELL_CASE(2, 4) STFEM_CELL_CASE(3, 6) STFEM_CELL_CASE(4, 8) // h-transfers
  STFEM_CELL_CASE(1, 3) STFEM_CELL_CASE(1, 4) STFEM_CELL_CASE(2, 3) STFEM_CELL_CASE(3, 4) // (p-transfers on the same cells)
#undef STFEM_CELL_CASE
  return 1;
}

// the y and z passes of the prolongation can run as one kernel: same coarse degree, refinement and embedding matrix along both
static bool fuse_yz = [] {
  const char *e = getenv("STFEM_TRANSFER_FUSE");
  return !e || atoi(e) != 0;
}();

// out (dims of `to`) (+)= (B2 (x) B1 (x) B0) in; order: the axes in `order`, smallest intermediates first
// cell: 0 = table-driven passes only (interpolation), 1 = prolongation, 2 = restriction in cell form along y and z
template <typename T>
int apply3(stfem_transfer *t, const Band B[3], void *out, const void *in, const int order[3], int add, int cell, hipStream_t s)
{
  int dims[3] = {B[0].n_in, B[1].n_in, B[2].n_in};
  const T *cur = static_cast<const T *>(in);
  // (measured on the cfg-1 levels, two Q4 blocks: fp32 h 0.231 -> 0.153 ms, fp32 / fp64 p 0.208 -> 0.172 / 0.315 -> 0.253 ms; fp64 h with 81 weights in
  // scalar registers 0.332 -> 0.340 ms, and below ~150 000 threads the one-cell-per-thread form has too few of them: both keep the three passes)
  const long long yz_threads = (long long)B[0].n_out * t->ncc[1] * t->ncc[2];
  const bool heavy = sizeof(T) == 8 && t->pc[1] == 4 && t->Rn[1] == 8;
  if (cell == 1 && fuse_yz && order[0] == 0 && t->pc[1] == t->pc[2] && t->Rn[1] == t->Rn[2] && t->Rn[1] > t->pc[1] && yz_threads >= 150000 && !heavy &&
      std::equal(t->L[1], t->L[1] + 11 * 6, t->L[2])) { // x pass (tables), then y and z in one kernel
    dims[0] = B[0].n_out;
    T *mid = static_cast<T *>(t->d_tmp[0]);
    int st = launch_axis<T>(mid, cur, dims, 0, B[0], 0, s);
    if (st != STFEM_OK) return st;
    st = launch_cell_yz<T>(t->pc[1], t->Rn[1], static_cast<T *>(out), mid, dims[0], t->ncc[1], t->ncc[2], t->L[1], t->flags[1], t->flags[2], add, s);
    if (st <= 0) return st;
    dims[0] = B[0].n_in; // no instantiation: the three-pass form below
  }
  for (int step = 0; step < 3; ++step) {
    const int ax = order[step];
    dims[ax] = B[ax].n_out;
    T *dst = step == 2 ? static_cast<T *>(out) : static_cast<T *>(t->d_tmp[step]);
    int st = 1;
    if (cell && ax > 0 && t->Rn[ax] > t->pc[ax] && !(cell == 2 && ax == 2 && !t->cell_restrict_z)) { // (an axis with the same cells and degree on both levels is a copy: table-driven)
      const long long S = ax == 1 ? dims[0] : (long long)dims[0] * dims[1];
      const long long total = S * t->ncc[ax] * (ax == 1 ? dims[2] : 1);
      st = launch_cell<T>(cell == 1, t->pc[ax], t->Rn[ax], dst, cur, S, t->ncc[ax], total, t->L[ax], t->flags[ax], step == 2 ? add : 0, s);
      if (st < 0) return st;
    }
    if (st == 1) st = launch_axis<T>(dst, cur, dims, ax, B[ax], step == 2 ? add : 0, s);
    if (st != STFEM_OK) return st;
    cur = dst;
  }
  return STFEM_OK;
}

int run(stfem_transfer *t, const Band B[3], stfem_ctx *to, stfem_ctx *from, stfem_vec *dst, const stfem_vec *src, bool expanding, int add,
        void *stream, int cell = 0)
{
  if (!t || !dst || !src) return STFEM_ERR_INVALID_ARGUMENT;
  if (dst->ctx != to || src->ctx != from || dst->nb != src->nb) return STFEM_ERR_SHAPE_MISMATCH;
  // expanding (prolongation): x, y, z keeps the intermediates small; contracting: z, y, x
  const int up[3] = {0, 1, 2}, down[3] = {2, 1, 0};
  hipStream_t s = static_cast<hipStream_t>(stream);
  for (int b = 0; b < dst->nb; ++b) {
    const int st = to->prec == 0 ? apply3<double>(t, B, dst->blk[b], src->blk[b], expanding ? up : down, add, cell, s)
                                 : apply3<float>(t, B, dst->blk[b], src->blk[b], expanding ? up : down, add, cell, s);
    if (st != STFEM_OK) return st;
  }
  return STFEM_OK;
}

} // namespace

extern "C" {

const char *stfem_transfer_last_error(void) { return g_transfer_err; }

int stfem_transfer_create(stfem_ctx *fine, stfem_ctx *coarse, stfem_transfer **out)
{
  return stfem_transfer_create_partitioned(fine, coarse, 0, out);
}

int stfem_transfer_create_partitioned(stfem_ctx *fine, stfem_ctx *coarse, int neighbour_mask, stfem_transfer **out)
{
  if (!fine || !coarse || !out || (neighbour_mask & ~48) || (neighbour_mask & (fine->dmask | coarse->dmask))) return STFEM_ERR_INVALID_ARGUMENT;
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
  if (const char *e = getenv("STFEM_TRANSFER_TABLES")) t->cell_form = atoi(e) == 0;
  for (int d = 0; d < 3; ++d) {
    std::vector<double> P, I;
    line_matrices(fine->nc[d], fine->p, coarse->nc[d], coarse->p, P, I);
    const std::vector<double> P0 = P;
    const int n_f = fine->nd[d], n_c = coarse->nd[d];
    // zero-boundary constraints of both levels: constrained rows are not written, constrained columns read as 0
    auto constrained = [&](const stfem_ctx *c, int i, int n) { return (i == 0 && (c->dmask >> (2 * d) & 1)) || (i == n - 1 && (c->dmask >> (2 * d + 1) & 1)); };
    // a slab with a neighbour above: its top fine plane is the ghost copy of the neighbour's bottom plane (stfem.h: halo support) and
    // is restricted THERE; here it does not contribute, and the add-exchange of the coarse interface planes completes the sums
    const bool ghost_top = d == 2 && (neighbour_mask & 32);
    std::vector<double> R(size_t(n_c) * n_f);
    for (int f = 0; f < n_f; ++f)
      for (int c = 0; c < n_c; ++c) {
        if (constrained(fine, f, n_f) || constrained(coarse, c, n_c)) P[size_t(f) * n_c + c] = 0.0, I[size_t(c) * n_f + f] = 0.0;
        R[size_t(c) * n_f + f] = (ghost_top && f == n_f - 1) ? 0.0 : P[size_t(f) * n_c + c];
      }
    if (ghost_top) t->cell_restrict_z = false; // (the cell form of the restriction has no such mask: table-driven pass along z)
    // cell form: the block of cell 0 of the unconstrained embedding (the same in every cell)
    t->pc[d] = coarse->p;
    t->Rn[d] = (fine->nc[d] / coarse->nc[d]) * fine->p;
    t->ncc[d] = coarse->nc[d];
    t->flags[d] = (constrained(coarse, 0, n_c) ? CF_LO_C : 0) | (constrained(coarse, n_c - 1, n_c) ? CF_HI_C : 0) |
                  (constrained(fine, 0, n_f) ? CF_LO_F : 0) | (constrained(fine, n_f - 1, n_f) ? CF_HI_F : 0);
    for (int j = 0; j <= t->Rn[d]; ++j)
      for (int a = 0; a <= t->pc[d]; ++a) t->L[d][j * (t->pc[d] + 1) + a] = P0[size_t(j) * n_c + a];
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
  return t ? run(t, t->P, t->fine, t->coarse, dst_fine, src_coarse, true, add, stream, t->cell_form ? 1 : 0) : STFEM_ERR_INVALID_ARGUMENT;
}
int stfem_transfer_restrict(stfem_transfer *t, stfem_vec *dst_coarse, const stfem_vec *src_fine, int add, void *stream)
{
  return t ? run(t, t->R, t->coarse, t->fine, dst_coarse, src_fine, false, add, stream, t->cell_form ? 2 : 0) : STFEM_ERR_INVALID_ARGUMENT;
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

// ---- stream capture: a fixed sequence of launches (one V-cycle: ~1300 kernels, most of them on coarse levels where the
// launch costs more than the kernel) recorded once into a hipGraph and replayed
struct stfem_graph {
  hipGraphExec_t exec = nullptr;
};

int stfem_stream_create(void **stream_out)
{
  if (!stream_out) return STFEM_ERR_INVALID_ARGUMENT;
  hipStream_t s = nullptr;
  TR_TRY(hipStreamCreate(&s)); // a blocking stream: ordered against the legacy default stream the other calls use
  *stream_out = s;
  return STFEM_OK;
}
void stfem_stream_destroy(void *stream)
{
  if (stream) (void)hipStreamDestroy(static_cast<hipStream_t>(stream));
}
int stfem_stream_synchronize(void *stream)
{
  TR_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  return STFEM_OK;
}
int stfem_graph_begin(void *stream)
{
  if (!stream) return STFEM_ERR_INVALID_ARGUMENT; // the legacy default stream cannot be captured
  TR_TRY(hipStreamBeginCapture(static_cast<hipStream_t>(stream), hipStreamCaptureModeThreadLocal));
  return STFEM_OK;
}
int stfem_graph_end(void *stream, stfem_graph **out)
{
  if (!stream || !out) return STFEM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  hipGraph_t g = nullptr;
  TR_TRY(hipStreamEndCapture(static_cast<hipStream_t>(stream), &g));
  stfem_graph *r = new stfem_graph;
  const hipError_t e = hipGraphInstantiate(&r->exec, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) {
    snprintf(g_transfer_err, sizeof(g_transfer_err), "hipGraphInstantiate: %s", hipGetErrorString(e));
    delete r;
    return STFEM_ERR_HIP;
  }
  *out = r;
  return STFEM_OK;
}
int stfem_graph_launch(stfem_graph *g, void *stream)
{
  if (!g || !g->exec) return STFEM_ERR_INVALID_ARGUMENT;
  TR_TRY(hipGraphLaunch(g->exec, static_cast<hipStream_t>(stream)));
  return STFEM_OK;
}
void stfem_graph_destroy(stfem_graph *g)
{
  if (!g) return;
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  delete g;
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
