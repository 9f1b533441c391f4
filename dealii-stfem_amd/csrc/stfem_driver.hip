// What the slab driver needs around the operator: load vectors, nodal interpolation points,
// error norms and a block axpby.  None of it is on the vmult hot path; the kernels are plain (one workgroup
// per cell, the MappingQ1 Jacobian from the eight cell vertices on the fly).
//
// Replaces, for the structured meshes of this library:
//   VectorTools::create_right_hand_side(mapping, dof_handler, quad, f, rhs, constraints)  (tests/tp_01.cc:382-392)
//       -> stfem_quadrature_points + stfem_integrate_rhs (the caller evaluates f at the points)
//   VectorTools::interpolate(mapping, dof_handler, u, vec)                                (tests/tp_01.cc:393-400)
//       -> stfem_support_points + stfem_vector_upload
//   VectorTools::integrate_difference (L2, Linfty, H1-seminorm) of ErrorCalculator::evaluate_error
//       (include/exact_solution.h:534-600)                                                -> stfem_integrate_difference
//   the vector arithmetic of SolverFGMRES / TimeIntegrator (add, sadd, equ)               -> stfem_vector_axpby
#include "stfem_internal.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {

thread_local char g_driver_err[256] = "";
#define DRV_TRY(call)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      snprintf(g_driver_err, sizeof(g_driver_err), "%s: %s", #call, hipGetErrorString(e_)); \
      return STFEM_ERR_HIP;                                                                 \
    }                                                                                       \
  } while (0)

// vertices of the structured block on the host (the context keeps them for general meshes only)
std::vector<double> host_vertices(const stfem_ctx *c)
{
  if (!c->vertices.empty()) return c->vertices;
  std::vector<double> v(size_t(c->nc[0] + 1) * (c->nc[1] + 1) * (c->nc[2] + 1) * 3);
  size_t o = 0;
  for (int k = 0; k <= c->nc[2]; ++k)
    for (int j = 0; j <= c->nc[1]; ++j)
      for (int i = 0; i <= c->nc[0]; ++i, ++o) {
        v[3 * o] = c->lower[0] + c->h[0] * i;
        v[3 * o + 1] = c->lower[1] + c->h[1] * j;
        v[3 * o + 2] = c->lower[2] + c->h[2] * k;
      }
  return v;
}

// trilinear map of the reference point xi in cell (cx, cy, cz)
void map_point(const stfem_ctx *c, const std::vector<double> &v, int cx, int cy, int cz, const double xi[3], double out[3])
{
  const size_t nvx = c->nc[0] + 1, nvy = c->nc[1] + 1;
  out[0] = out[1] = out[2] = 0.0;
  for (int k = 0; k < 2; ++k)
    for (int j = 0; j < 2; ++j)
      for (int i = 0; i < 2; ++i) {
        const double w = (i ? xi[0] : 1 - xi[0]) * (j ? xi[1] : 1 - xi[1]) * (k ? xi[2] : 1 - xi[2]);
        const double *X = v.data() + 3 * ((cx + i) + nvx * ((cy + j) + nvy * size_t(cz + k)));
        for (int d = 0; d < 3; ++d) out[d] += w * X[d];
      }
}

struct CellGeomParams {
  const double *vertices; // device
  int ncx, ncy, ncz, nx, ny, p, nq, dmask;
  const double *S, *D;    // [nq][n]: nodal basis and its reference derivative at the quadrature points (device)
  const double *xq, *wq;  // [nq] (device)
};

// Jacobian of the trilinear map at the reference point: J[d][e] = d x_d / d xi_e
__device__ void jacobian(const CellGeomParams &g, int cx, int cy, int cz, const double xi[3], double J[3][3])
{
  const long long nvx = g.ncx + 1, nvy = g.ncy + 1;
  for (int d = 0; d < 3; ++d)
    for (int e = 0; e < 3; ++e) J[d][e] = 0.0;
  for (int k = 0; k < 2; ++k)
    for (int j = 0; j < 2; ++j)
      for (int i = 0; i < 2; ++i) {
        const double *X = g.vertices + 3 * ((cx + i) + nvx * ((cy + j) + nvy * (long long)(cz + k)));
        const double fx = i ? xi[0] : 1 - xi[0], fy = j ? xi[1] : 1 - xi[1], fz = k ? xi[2] : 1 - xi[2];
        const double dx = i ? 1.0 : -1.0, dy = j ? 1.0 : -1.0, dz = k ? 1.0 : -1.0;
        for (int d = 0; d < 3; ++d) {
          J[d][0] += X[d] * dx * fy * fz;
          J[d][1] += X[d] * fx * dy * fz;
          J[d][2] += X[d] * fx * fy * dz;
        }
      }
}
__device__ double det3(const double J[3][3])
{
  return J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
         J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
}

// the trilinear map of the reference point xi in cell (cx, cy, cz)
__device__ void map_point_dev(const CellGeomParams &g, int cx, int cy, int cz, const double xi[3], double x[3])
{
  const long long nvx = g.ncx + 1, nvy = g.ncy + 1;
  x[0] = x[1] = x[2] = 0.0;
  for (int k = 0; k < 2; ++k)
    for (int j = 0; j < 2; ++j)
      for (int i = 0; i < 2; ++i) {
        const double w = (i ? xi[0] : 1 - xi[0]) * (j ? xi[1] : 1 - xi[1]) * (k ? xi[2] : 1 - xi[2]);
        const double *V = g.vertices + 3 * ((cx + i) + nvx * ((cy + j) + nvy * (long long)(cz + k)));
        for (int d = 0; d < 3; ++d) x[d] += w * V[d];
      }
}

// f(x) = amplitude * prod_d sin(2 pi frequency x_d): the separable functions of the reference's convergence tests (exact solutions and
// right-hand sides of include/exact_solution.h:27-81, 147-197 at a fixed time), evaluated on the device instead of handed in point by point
struct ProductFn {
  int on;
  double amplitude, frequency;
};
__device__ double product_value(const ProductFn &f, const double x[3], double grad[3])
{
  const double w = 6.283185307179586476925286766559 * f.frequency;
  const double s[3] = {sin(w * x[0]), sin(w * x[1]), sin(w * x[2])}, c[3] = {cos(w * x[0]), cos(w * x[1]), cos(w * x[2])};
  if (grad) {
    grad[0] = f.amplitude * w * c[0] * s[1] * s[2];
    grad[1] = f.amplitude * w * s[0] * c[1] * s[2];
    grad[2] = f.amplitude * w * s[0] * s[1] * c[2];
  }
  return f.amplitude * s[0] * s[1] * s[2];
}

// rhs_a += sum_q JxW_q f_q phi_a(x_q); one workgroup per cell; constrained rows stay 0
template <typename T>
__global__ __launch_bounds__(256) void integrate_rhs_kernel(const CellGeomParams g, const double *__restrict__ fq, T *__restrict__ dst, const ProductFn pf)
{
  extern __shared__ double sm[]; // [nq^3] JxW f
  const int n = g.p + 1, nq = g.nq, nq3 = nq * nq * nq, nloc = n * n * n;
  const long long cell = blockIdx.x;
  const int cx = int(cell % g.ncx), cy = int((cell / g.ncx) % g.ncy), cz = int(cell / ((long long)g.ncx * g.ncy));
  for (int q = threadIdx.x; q < nq3; q += 256) {
    const int qx = q % nq, qy = (q / nq) % nq, qz = q / (nq * nq);
    const double xi[3] = {g.xq[qx], g.xq[qy], g.xq[qz]};
    double J[3][3];
    jacobian(g, cx, cy, cz, xi, J);
    double fv;
    if (pf.on) {
      double x[3];
      map_point_dev(g, cx, cy, cz, xi, x);
      fv = product_value(pf, x, nullptr);
    } else fv = fq[cell * nq3 + q];
    sm[q] = det3(J) * g.wq[qx] * g.wq[qy] * g.wq[qz] * fv;
  }
  __syncthreads();
  for (int a = threadIdx.x; a < nloc; a += 256) {
    const int ax = a % n, ay = (a / n) % n, az = a / (n * n);
    const int ix = g.p * cx + ax, iy = g.p * cy + ay, iz = g.p * cz + az;
    const bool con = ((g.dmask & 1) && ix == 0) || ((g.dmask & 2) && ix == g.nx - 1) || ((g.dmask & 4) && iy == 0) ||
                     ((g.dmask & 8) && iy == g.ny - 1) || ((g.dmask & 16) && iz == 0) || ((g.dmask & 32) && iz == g.p * g.ncz);
    if (con) continue;
    double s = 0.0;
    for (int qz = 0; qz < nq; ++qz)
      for (int qy = 0; qy < nq; ++qy) {
        const double syz = g.S[qy * n + ay] * g.S[qz * n + az];
        for (int qx = 0; qx < nq; ++qx) s += sm[qx + nq * (qy + nq * qz)] * g.S[qx * n + ax] * syz;
      }
    atomicAdd(dst + ix + (long long)g.nx * (iy + (long long)g.ny * iz), T(s));
  }
}

// per cell: sum_q JxW (u_h - u)^2, max_q |u_h - u|, sum_q JxW |grad u_h - grad u|^2  -> out[cell][3]
template <typename T>
__global__ __launch_bounds__(256) void integrate_difference_kernel(const CellGeomParams g, const T *__restrict__ u,
                                                                   const double *__restrict__ exact, const double *__restrict__ exact_grad,
                                                                   double *__restrict__ out, const ProductFn pf)
{
  extern __shared__ double sm[]; // [nloc] cell values, then 3 x 256 reduction
  const int n = g.p + 1, nq = g.nq, nq3 = nq * nq * nq, nloc = n * n * n;
  double *ul = sm, *red = sm + nloc;
  const long long cell = blockIdx.x;
  const int cx = int(cell % g.ncx), cy = int((cell / g.ncx) % g.ncy), cz = int(cell / ((long long)g.ncx * g.ncy));
  for (int a = threadIdx.x; a < nloc; a += 256) {
    const int ax = a % n, ay = (a / n) % n, az = a / (n * n);
    ul[a] = double(u[(g.p * cx + ax) + (long long)g.nx * ((g.p * cy + ay) + (long long)g.ny * (g.p * cz + az))]);
  }
  __syncthreads();
  double l2 = 0.0, l8 = 0.0, h1 = 0.0;
  for (int q = threadIdx.x; q < nq3; q += 256) {
    const int qx = q % nq, qy = (q / nq) % nq, qz = q / (nq * nq);
    const double xi[3] = {g.xq[qx], g.xq[qy], g.xq[qz]};
    double J[3][3];
    jacobian(g, cx, cy, cz, xi, J);
    const double det = det3(J), JxW = det * g.wq[qx] * g.wq[qy] * g.wq[qz];
    double val = 0.0, gr[3] = {0, 0, 0};
    for (int az = 0; az < n; ++az)
      for (int ay = 0; ay < n; ++ay)
        for (int ax = 0; ax < n; ++ax) {
          const double w = ul[ax + n * (ay + n * az)];
          const double sx = g.S[qx * n + ax], sy = g.S[qy * n + ay], sz = g.S[qz * n + az];
          val += w * sx * sy * sz;
          gr[0] += w * g.D[qx * n + ax] * sy * sz;
          gr[1] += w * sx * g.D[qy * n + ay] * sz;
          gr[2] += w * sx * sy * g.D[qz * n + az];
        }
    double ex, exg[3] = {0, 0, 0};
    if (pf.on) {
      double x[3];
      map_point_dev(g, cx, cy, cz, xi, x);
      ex = product_value(pf, x, exg);
    } else {
      ex = exact[cell * nq3 + q];
      if (exact_grad)
        for (int d = 0; d < 3; ++d) exg[d] = exact_grad[(cell * nq3 + q) * 3 + d];
    }
    const double e = val - ex;
    l2 += JxW * e * e;
    l8 = fmax(l8, fabs(e));
    if (exact_grad || pf.on) {
      // physical gradient = J^-T reference gradient
      const double id = 1.0 / det;
      double Ji[3][3];
      Ji[0][0] = (J[1][1] * J[2][2] - J[1][2] * J[2][1]) * id; Ji[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id;
      Ji[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id; Ji[1][0] = (J[1][2] * J[2][0] - J[1][0] * J[2][2]) * id;
      Ji[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id; Ji[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
      Ji[2][0] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) * id; Ji[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id;
      Ji[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
      for (int d = 0; d < 3; ++d) {
        const double gd = gr[0] * Ji[0][d] + gr[1] * Ji[1][d] + gr[2] * Ji[2][d] - exg[d];
        h1 += JxW * gd * gd;
      }
    }
  }
  red[threadIdx.x] = l2; red[256 + threadIdx.x] = l8; red[512 + threadIdx.x] = h1;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (int(threadIdx.x) < s) {
      red[threadIdx.x] += red[threadIdx.x + s];
      red[256 + threadIdx.x] = fmax(red[256 + threadIdx.x], red[256 + threadIdx.x + s]);
      red[512 + threadIdx.x] += red[512 + threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[cell * 3] = red[0]; out[cell * 3 + 1] = red[256]; out[cell * 3 + 2] = red[512]; }
}

// y = a x + b y on up to eight blocks per launch (blockIdx.y = block).  A zero factor means "not read": a = 0 never
// touches x, b = 0 never touches y's old content (0 * NaN and 0 * Inf would survive otherwise, where deal.II's
// `dst = 0.` / equ() assign); x and y may be the same vector (no __restrict__).
struct AxpbyBlocks {
  const void *x[8];
  void *y[8];
};
template <typename T>
__global__ __launch_bounds__(256) void axpby_kernel(int64_t n, T a, T b, const AxpbyBlocks blocks)
{
  const T *x = static_cast<const T *>(blocks.x[blockIdx.y]);
  T *y = static_cast<T *>(blocks.y[blockIdx.y]);
  const int64_t i0 = int64_t(blockIdx.x) * blockDim.x + threadIdx.x, stride = int64_t(gridDim.x) * blockDim.x;
  if (a == T(0) && b == T(0))
    for (int64_t i = i0; i < n; i += stride) y[i] = T(0);
  else if (a == T(0))
    for (int64_t i = i0; i < n; i += stride) y[i] = b * y[i];
  else if (b == T(0))
    for (int64_t i = i0; i < n; i += stride) y[i] = a * x[i];
  else
    for (int64_t i = i0; i < n; i += stride) y[i] = a * x[i] + b * y[i];
}

// the same on arrays of different lengths in one launch (the blocks of a two-variable vector: velocity and pressure blocks)
struct AxpbyMany {
  const void *x[8];
  void *y[8];
  long long len[8];
};
template <typename T>
__global__ __launch_bounds__(256) void axpby_many_kernel(T a, T b, const AxpbyMany v)
{
  const T *x = static_cast<const T *>(v.x[blockIdx.y]);
  T *y = static_cast<T *>(v.y[blockIdx.y]);
  const int64_t n = v.len[blockIdx.y];
  const int64_t i0 = int64_t(blockIdx.x) * blockDim.x + threadIdx.x, stride = int64_t(gridDim.x) * blockDim.x;
  if (a == T(0) && b == T(0))
    for (int64_t i = i0; i < n; i += stride) y[i] = T(0);
  else if (a == T(0))
    for (int64_t i = i0; i < n; i += stride) y[i] = b * y[i];
  else if (b == T(0))
    for (int64_t i = i0; i < n; i += stride) y[i] = a * x[i];
  else
    for (int64_t i = i0; i < n; i += stride) y[i] = a * x[i] + b * y[i];
}

// device copies of the geometry and of the tables of QGauss(nq) against the context's nodal basis
struct GeomUpload {
  double *d = nullptr;
  CellGeomParams g{};
  int build(const stfem_ctx *c, int nq)
  {
    const int n = c->p + 1;
    const std::vector<double> v = host_vertices(c);
    std::vector<double> xq, wq;
    stfem::gauss_rule(nq, xq, wq);
    stfem::Mat S, D;
    stfem::lagrange_tables(c->tab.nodes, xq, S, D);
    std::vector<double> all(v);
    const size_t oS = all.size();
    all.insert(all.end(), S.begin(), S.end());
    const size_t oD = all.size();
    all.insert(all.end(), D.begin(), D.end());
    const size_t ox = all.size();
    all.insert(all.end(), xq.begin(), xq.end());
    const size_t ow = all.size();
    all.insert(all.end(), wq.begin(), wq.end());
    if (hipMalloc(&d, all.size() * sizeof(double)) != hipSuccess) return STFEM_ERR_OUT_OF_MEMORY;
    if (hipMemcpy(d, all.data(), all.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return STFEM_ERR_HIP;
    g.vertices = d; g.S = d + oS; g.D = d + oD; g.xq = d + ox; g.wq = d + ow;
    g.ncx = c->nc[0]; g.ncy = c->nc[1]; g.ncz = c->nc[2]; g.nx = c->nd[0]; g.ny = c->nd[1];
    g.p = c->p; g.nq = nq; g.dmask = c->dmask;
    (void)n;
    return STFEM_OK;
  }
  ~GeomUpload()
  {
    if (d) (void)hipFree(d);
  }
};

} // namespace

extern "C" {

const char *stfem_driver_last_error(void) { return g_driver_err; }

int stfem_support_points(const stfem_ctx *c, double *out)
{
  if (!c || !out) return STFEM_ERR_INVALID_ARGUMENT;
  const std::vector<double> v = host_vertices(c);
  const int p = c->p;
  for (int iz = 0; iz < c->nd[2]; ++iz)
    for (int iy = 0; iy < c->nd[1]; ++iy)
      for (int ix = 0; ix < c->nd[0]; ++ix) {
        const int cx = std::min(ix / p, c->nc[0] - 1), cy = std::min(iy / p, c->nc[1] - 1), cz = std::min(iz / p, c->nc[2] - 1);
        const double xi[3] = {c->tab.nodes[ix - p * cx], c->tab.nodes[iy - p * cy], c->tab.nodes[iz - p * cz]};
        map_point(c, v, cx, cy, cz, xi, out + 3 * (ix + size_t(c->nd[0]) * (iy + size_t(c->nd[1]) * iz)));
      }
  return STFEM_OK;
}

int stfem_quadrature_points(const stfem_ctx *c, int nq, double *out)
{
  if (!c || !out || nq < 1 || nq > 8) return STFEM_ERR_INVALID_ARGUMENT;
  const std::vector<double> v = host_vertices(c);
  std::vector<double> xq, wq;
  stfem::gauss_rule(nq, xq, wq);
  size_t o = 0;
  for (int cz = 0; cz < c->nc[2]; ++cz)
    for (int cy = 0; cy < c->nc[1]; ++cy)
      for (int cx = 0; cx < c->nc[0]; ++cx)
        for (int qz = 0; qz < nq; ++qz)
          for (int qy = 0; qy < nq; ++qy)
            for (int qx = 0; qx < nq; ++qx, ++o) {
              const double xi[3] = {xq[qx], xq[qy], xq[qz]};
              map_point(c, v, cx, cy, cz, xi, out + 3 * o);
            }
  return STFEM_OK;
}

static int integrate_rhs_impl(stfem_ctx *c, int nq, const double *f_at_points, const ProductFn pf, stfem_vec *dst, int block, void *stream)
{
  if (!c || (!f_at_points && !pf.on) || !dst || dst->ctx != c || block < 0 || block >= dst->nb || nq < 1 || nq > 8) return STFEM_ERR_INVALID_ARGUMENT;
  DRV_TRY(hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  GeomUpload geo;
  int rc = geo.build(c, nq);
  if (rc != STFEM_OK) return rc;
  const size_t nf = size_t(c->ncells) * nq * nq * nq;
  double *d_f = nullptr;
  hipError_t e = hipSuccess;
  if (!pf.on) {
    if (hipMalloc(&d_f, nf * sizeof(double)) != hipSuccess) return STFEM_ERR_OUT_OF_MEMORY;
    e = hipMemcpyAsync(d_f, f_at_points, nf * sizeof(double), hipMemcpyHostToDevice, st);
  }
  if (e == hipSuccess) e = hipMemsetAsync(dst->blk[block], 0, size_t(c->ndofs) * c->es, st);
  if (e == hipSuccess) {
    const size_t lds = size_t(nq) * nq * nq * sizeof(double);
    if (c->prec)
      hipLaunchKernelGGL(integrate_rhs_kernel<float>, dim3((unsigned)c->ncells), dim3(256), lds, st, geo.g, d_f, static_cast<float *>(dst->blk[block]), pf);
    else
      hipLaunchKernelGGL(integrate_rhs_kernel<double>, dim3((unsigned)c->ncells), dim3(256), lds, st, geo.g, d_f, static_cast<double *>(dst->blk[block]), pf);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (d_f) (void)hipFree(d_f);
  if (e != hipSuccess) {
    snprintf(g_driver_err, sizeof(g_driver_err), "stfem_integrate_rhs: %s", hipGetErrorString(e));
    return STFEM_ERR_HIP;
  }
  return STFEM_OK;
}

int stfem_integrate_rhs(stfem_ctx *c, int nq, const double *f_at_points, stfem_vec *dst, int block, void *stream)
{
  return integrate_rhs_impl(c, nq, f_at_points, ProductFn{0, 0.0, 0.0}, dst, block, stream);
}
int stfem_integrate_rhs_product(stfem_ctx *c, int nq, double amplitude, double frequency, stfem_vec *dst, int block, void *stream)
{
  return integrate_rhs_impl(c, nq, nullptr, ProductFn{1, amplitude, frequency}, dst, block, stream);
}

static int integrate_difference_impl(stfem_ctx *c, int nq, const stfem_vec *u, int block, const double *exact_at_points,
                                     const double *exact_grad_at_points, const ProductFn pf, double out[3], void *stream)
{
  if (!c || !u || u->ctx != c || block < 0 || block >= u->nb || (!exact_at_points && !pf.on) || !out || nq < 1 || nq > 8)
    return STFEM_ERR_INVALID_ARGUMENT;
  DRV_TRY(hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  GeomUpload geo;
  int rc = geo.build(c, nq);
  if (rc != STFEM_OK) return rc;
  const size_t npts = size_t(c->ncells) * nq * nq * nq;
  double *d_e = nullptr, *d_g = nullptr, *d_out = nullptr;
  auto cleanup = [&]() {
    if (d_e) (void)hipFree(d_e);
    if (d_g) (void)hipFree(d_g);
    if (d_out) (void)hipFree(d_out);
  };
  if ((!pf.on && hipMalloc(&d_e, npts * sizeof(double)) != hipSuccess) || hipMalloc(&d_out, size_t(c->ncells) * 3 * sizeof(double)) != hipSuccess ||
      (!pf.on && exact_grad_at_points && hipMalloc(&d_g, npts * 3 * sizeof(double)) != hipSuccess)) {
    cleanup();
    return STFEM_ERR_OUT_OF_MEMORY;
  }
  hipError_t e = hipSuccess;
  if (!pf.on) e = hipMemcpyAsync(d_e, exact_at_points, npts * sizeof(double), hipMemcpyHostToDevice, st);
  if (e == hipSuccess && !pf.on && exact_grad_at_points) e = hipMemcpyAsync(d_g, exact_grad_at_points, npts * 3 * sizeof(double), hipMemcpyHostToDevice, st);
  std::vector<double> part(size_t(c->ncells) * 3);
  if (e == hipSuccess) {
    const int n = c->p + 1;
    const size_t lds = (size_t(n) * n * n + 3 * 256) * sizeof(double);
    if (c->prec)
      hipLaunchKernelGGL(integrate_difference_kernel<float>, dim3((unsigned)c->ncells), dim3(256), lds, st, geo.g,
                         static_cast<const float *>(u->blk[block]), d_e, d_g, d_out, pf);
    else
      hipLaunchKernelGGL(integrate_difference_kernel<double>, dim3((unsigned)c->ncells), dim3(256), lds, st, geo.g,
                         static_cast<const double *>(u->blk[block]), d_e, d_g, d_out, pf);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(part.data(), d_out, part.size() * sizeof(double), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  cleanup();
  if (e != hipSuccess) {
    snprintf(g_driver_err, sizeof(g_driver_err), "stfem_integrate_difference: %s", hipGetErrorString(e));
    return STFEM_ERR_HIP;
  }
  out[0] = out[1] = out[2] = 0.0;
  for (int64_t cell = 0; cell < c->ncells; ++cell) {
    out[0] += part[cell * 3];
    out[1] = std::max(out[1], part[cell * 3 + 1]);
    out[2] += part[cell * 3 + 2];
  }
  return STFEM_OK;
}

int stfem_integrate_difference(stfem_ctx *c, int nq, const stfem_vec *u, int block, const double *exact_at_points,
                               const double *exact_grad_at_points, double out[3], void *stream)
{
  return integrate_difference_impl(c, nq, u, block, exact_at_points, exact_grad_at_points, ProductFn{0, 0.0, 0.0}, out, stream);
}
int stfem_integrate_difference_product(stfem_ctx *c, int nq, const stfem_vec *u, int block, double amplitude, double frequency, double out[3],
                                       void *stream)
{
  return integrate_difference_impl(c, nq, u, block, nullptr, nullptr, ProductFn{1, amplitude, frequency}, out, stream);
}

int stfem_gauss_rule(int n, double *points, double *weights)
{
  if (n < 1 || n > 16 || !points || !weights) return STFEM_ERR_INVALID_ARGUMENT;
  std::vector<double> x, w;
  stfem::gauss_rule(n, x, w);
  std::copy(x.begin(), x.end(), points);
  std::copy(w.begin(), w.end(), weights);
  return STFEM_OK;
}

int stfem_fe_time_points(int type, int r, double *points)
{
  if ((type != 0 && type != 1) || r < 0 || r > 8 || !points) return STFEM_ERR_INVALID_ARGUMENT;
  // get_time_quad (fe_time.cc:152-161): QGaussLobatto(r + 1) for cG(r), QGaussRadau(r + 1, right) for dG(r)
  if (type == 0 && r < 1) return STFEM_ERR_INVALID_ARGUMENT;
  const std::vector<double> x = type == 0 ? stfem::lobatto_points(r + 1) : stfem::radau_right_points(r + 1);
  std::copy(x.begin(), x.end(), points);
  return STFEM_OK;
}

int stfem_vector_axpby(stfem_ctx *c, double a, const stfem_vec *x, double b, stfem_vec *y, void *stream)
{
  if (!c || !x || !y || x->ctx != c || y->ctx != c || x->nb != y->nb) return STFEM_ERR_INVALID_ARGUMENT;
  DRV_TRY(hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned grid = (unsigned)std::min<int64_t>((c->ndofs + 255) / 256, 4096);
  (void)hipGetLastError();
  for (int b0 = 0; b0 < x->nb; b0 += 8) {
    const int nb = std::min(8, x->nb - b0);
    AxpbyBlocks bl{};
    for (int j = 0; j < nb; ++j) {
      bl.x[j] = x->blk[b0 + j];
      bl.y[j] = y->blk[b0 + j];
    }
    if (c->prec) hipLaunchKernelGGL(axpby_kernel<float>, dim3(grid, nb), dim3(256), 0, st, c->ndofs, float(a), float(b), bl);
    else hipLaunchKernelGGL(axpby_kernel<double>, dim3(grid, nb), dim3(256), 0, st, c->ndofs, a, b, bl);
  }
  return hipGetLastError() == hipSuccess ? STFEM_OK : STFEM_ERR_HIP;
}

int stfem_axpby_many(stfem_ctx *c, int n_arrays, const int64_t *len, double a, const void *const *x, double b, void *const *y, void *stream)
{
  if (!c || n_arrays < 0 || (n_arrays > 0 && (!len || !y || (a != 0.0 && !x)))) return STFEM_ERR_INVALID_ARGUMENT;
  for (int j = 0; j < n_arrays; ++j)
    if (len[j] < 0 || !y[j] || (a != 0.0 && !x[j])) return STFEM_ERR_INVALID_ARGUMENT;
  DRV_TRY(hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  (void)hipGetLastError();
  for (int b0 = 0; b0 < n_arrays; b0 += 8) {
    const int nb = std::min(8, n_arrays - b0);
    AxpbyMany v{};
    long long longest = 0;
    for (int j = 0; j < nb; ++j) {
      v.x[j] = a != 0.0 ? x[b0 + j] : nullptr;
      v.y[j] = y[b0 + j];
      v.len[j] = len[b0 + j];
      longest = std::max<long long>(longest, len[b0 + j]);
    }
    if (longest == 0) continue;
    const unsigned grid = (unsigned)std::min<long long>((longest + 255) / 256, 4096);
    if (c->prec) hipLaunchKernelGGL(axpby_many_kernel<float>, dim3(grid, nb), dim3(256), 0, st, float(a), float(b), v);
    else hipLaunchKernelGGL(axpby_many_kernel<double>, dim3(grid, nb), dim3(256), 0, st, a, b, v);
  }
  return hipGetLastError() == hipSuccess ? STFEM_OK : STFEM_ERR_HIP;
}

int stfem_vector_set_zero(stfem_ctx *c, stfem_vec *y, void *stream)
{
  if (!c || !y || y->ctx != c) return STFEM_ERR_INVALID_ARGUMENT;
  DRV_TRY(hipSetDevice(c->device));
  const size_t bytes = size_t(c->ndofs) * (c->prec ? sizeof(float) : sizeof(double));
  for (int j = 0; j < y->nb; ++j) DRV_TRY(hipMemsetAsync(y->blk[j], 0, bytes, static_cast<hipStream_t>(stream)));
  return STFEM_OK;
}

} // extern "C"
