// Stokes two-field cell operator on MI355X (BASELINE configs[4]).
//
// Replaces, for the cell loop (LoopType::Cell: no weak boundary ids, delta0 = 0):
//   StokesMatrixFreeOperator::do_cell_integral_range / do_cell_integral_local
//       (reference include/operators.h:1501-1575, OperatorMode::none):
//       pressure.submit_value(div u); velocity.submit_gradient(nu grad u - p I)
//   the vector mass operator behind d/dt u (MatrixFreeOperator<dim, dim, Number>, operators.h:1135-1173)
//   SystemMatrixStokes::vmult -> tensorproduct_eval (operators.h:696-700, 825-867): per source time
//       dof one K.vmult + scatter with Alpha and one M.vmult + scatter with Beta.
// Here ONE launch per source time dof evaluates the cell once and scatters
//       wKu_j * (nu K u - B^T p) + wM_j * M u   into every velocity destination block j,
//       wKp_j * (div u, q)                       into every pressure destination block j
// with fp64 atomics (destinations zeroed first).  No CPU fallback.
//
// Thread layout: one wave owns two cells (32 lanes each, 27 = 3^3 active): in the evaluation
// phase a lane is a quadrature point, in the integration phase a velocity node (lanes 0..7 also
// a pressure node).  The 89 cell DoFs and the 13 flux values per point go through LDS.  The FE_Q(2)
// / FE_Q(1) shape values are products of 1D tables held in SGPRs (kernel arguments); the MappingQ1
// Jacobian is evaluated on the fly from the eight cell vertices (24 doubles per cell instead of a
// stored metric).  This is a first, correct version: it is not sum-factorised and not tuned.
#include "../../include/stfem.h"
#include "host_tables.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <new>
#include <vector>

namespace {

constexpr int MAXOUT = 8;

struct StokesParams {
  const double *vertices; // device, (nc+1)^3 * 3
  int ncx, ncy, ncz;
  int ndu[3], ndp[3];
  long long Nu, Np;
  int dmask;
  double nu;
  const double *u, *p; // source (u may be read with p == nullptr: mass only)
  int nout;
  double *out_u[MAXOUT], *out_p[MAXOUT];
  double wKu[MAXOUT], wKp[MAXOUT], wM[MAXOUT];
  double Su[9], Du[9], Sp[6]; // [q*3+a], [q*3+a], [q*2+a]
  double xq[3], wq[3];
};

__device__ __forceinline__ bool constrained_u(const StokesParams &prm, int ix, int iy, int iz)
{
  return ((prm.dmask & 1) && ix == 0) || ((prm.dmask & 2) && ix == prm.ndu[0] - 1) ||
         ((prm.dmask & 4) && iy == 0) || ((prm.dmask & 8) && iy == prm.ndu[1] - 1) ||
         ((prm.dmask & 16) && iz == 0) || ((prm.dmask & 32) && iz == prm.ndu[2] - 1);
}

// 256 threads = 4 waves = 8 cells
__global__ __launch_bounds__(256) void stokes_cell_kernel(const StokesParams prm)
{
  constexpr int CELL_LDS = 81 + 8 + 27 * 13; // u[3][27], p[8], per point: Fref[9], dq, mu[3]
  __shared__ double smem[8 * CELL_LDS];
  __shared__ double tS[9], tD[9], tP[6]; // 1D tables [q*3+a], [q*3+a], [q*2+a]
  if (threadIdx.x < 9) { tS[threadIdx.x] = prm.Su[threadIdx.x]; tD[threadIdx.x] = prm.Du[threadIdx.x]; }
  if (threadIdx.x < 6) tP[threadIdx.x] = prm.Sp[threadIdx.x];
  const int slot = threadIdx.x >> 5, t = threadIdx.x & 31;
  const long long ncells = (long long)prm.ncx * prm.ncy * prm.ncz;
  const long long cell = (long long)blockIdx.x * 8 + slot;
  const bool cell_ok = cell < ncells;
  const long long cc = cell_ok ? cell : 0;
  const int cx = int(cc % prm.ncx), cy = int((cc / prm.ncx) % prm.ncy), cz = int(cc / ((long long)prm.ncx * prm.ncy));
  double *ul = smem + slot * CELL_LDS, *pl = ul + 81, *fl = pl + 8;
  const bool active = cell_ok && t < 27;
  const int a = t % 3, b = (t / 3) % 3, c = t / 9; // node or quadrature point (x fastest)

  // ---- gather (read_dof_values: constrained velocity entries read as 0)
  const int ix = 2 * cx + a, iy = 2 * cy + b, iz = 2 * cz + c;
  const bool con = constrained_u(prm, ix, iy, iz);
  const long long gu = ix + (long long)prm.ndu[0] * (iy + (long long)prm.ndu[1] * iz);
  if (active) {
#pragma unroll
    for (int comp = 0; comp < 3; ++comp) ul[comp * 27 + t] = con ? 0.0 : prm.u[comp * prm.Nu + gu];
  }
  const int pa = t & 1, pb = (t >> 1) & 1, pc = (t >> 2) & 1;
  const long long gp = (cx + pa) + (long long)prm.ndp[0] * ((cy + pb) + (long long)prm.ndp[1] * (cz + pc));
  if (cell_ok && t < 8) pl[t] = prm.p ? prm.p[gp] : 0.0;
  __syncthreads();

  // ---- evaluate + quadrature-point operation (this lane = point (a, b, c))
  double Ji[3][3], JxW = 0.0;
  if (active) {
    const double x = prm.xq[a], y = prm.xq[b], z = prm.xq[c];
    const double fx[2] = {1 - x, x}, fy[2] = {1 - y, y}, fz[2] = {1 - z, z}, dd[2] = {-1.0, 1.0};
    double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    const long long nvx = prm.ncx + 1, nvy = prm.ncy + 1;
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const double *X = prm.vertices + 3 * ((cx + i) + nvx * ((cy + j) + nvy * (long long)(cz + k)));
#pragma unroll
          for (int d = 0; d < 3; ++d) {
            const double Xd = X[d];
            J[d][0] += Xd * dd[i] * fy[j] * fz[k];
            J[d][1] += Xd * fx[i] * dd[j] * fz[k];
            J[d][2] += Xd * fx[i] * fy[j] * dd[k];
          }
        }
    const double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
                       J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
    const double id = 1.0 / det;
    Ji[0][0] = (J[1][1] * J[2][2] - J[1][2] * J[2][1]) * id;
    Ji[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id;
    Ji[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
    Ji[1][0] = (J[1][2] * J[2][0] - J[1][0] * J[2][2]) * id;
    Ji[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id;
    Ji[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
    Ji[2][0] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) * id;
    Ji[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id;
    Ji[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
    JxW = det * prm.wq[a] * prm.wq[b] * prm.wq[c];

    // the node loops stay rolled (small code, few registers); 1D table rows come from LDS
    double sx[3], dx[3];
#pragma unroll
    for (int n = 0; n < 3; ++n) { sx[n] = tS[a * 3 + n]; dx[n] = tD[a * 3 + n]; }
    double gref[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, uval[3] = {0, 0, 0};
#pragma unroll 1
    for (int nc = 0; nc < 3; ++nc) {
      const double sz = tS[c * 3 + nc], dz = tD[c * 3 + nc];
#pragma unroll 1
      for (int nb = 0; nb < 3; ++nb) {
        const double sy = tS[b * 3 + nb], dy = tD[b * 3 + nb];
        const double syz = sy * sz, dyz = dy * sz, sdz = sy * dz;
#pragma unroll
        for (int na = 0; na < 3; ++na) {
          const int n = na + 3 * (nb + 3 * nc);
          const double gx = dx[na] * syz, gy = sx[na] * dyz, gz = sx[na] * sdz, val = sx[na] * syz;
#pragma unroll
          for (int comp = 0; comp < 3; ++comp) {
            const double w = ul[comp * 27 + n];
            gref[comp][0] = fma(w, gx, gref[comp][0]);
            gref[comp][1] = fma(w, gy, gref[comp][1]);
            gref[comp][2] = fma(w, gz, gref[comp][2]);
            uval[comp] = fma(w, val, uval[comp]);
          }
        }
      }
    }
    double pval = 0.0;
#pragma unroll
    for (int nc = 0; nc < 2; ++nc)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int na = 0; na < 2; ++na)
          pval = fma(pl[na + 2 * (nb + 2 * nc)], tP[a * 2 + na] * tP[b * 2 + nb] * tP[c * 2 + nc], pval);
    double grad[3][3];
#pragma unroll
    for (int comp = 0; comp < 3; ++comp)
#pragma unroll
      for (int d = 0; d < 3; ++d)
        grad[comp][d] = gref[comp][0] * Ji[0][d] + gref[comp][1] * Ji[1][d] + gref[comp][2] * Ji[2][d];
    const double divu = grad[0][0] + grad[1][1] + grad[2][2];
    // operators.h:1547-1553, 1570 (weights applied at scatter time)
#pragma unroll
    for (int comp = 0; comp < 3; ++comp) {
      double F[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) F[d] = (prm.nu * grad[comp][d] - (comp == d ? pval : 0.0)) * JxW;
#pragma unroll
      for (int e = 0; e < 3; ++e) fl[t * 13 + comp * 3 + e] = Ji[e][0] * F[0] + Ji[e][1] * F[1] + Ji[e][2] * F[2];
      fl[t * 13 + 10 + comp] = uval[comp] * JxW;
    }
    fl[t * 13 + 9] = divu * JxW;
  }
  __syncthreads();

  // ---- integrate (this lane = velocity node (a, b, c); lanes 0..7 also pressure node)
  double rK[3] = {0, 0, 0}, rM[3] = {0, 0, 0}, rP = 0.0;
  if (active) {
#pragma unroll 1
    for (int qc = 0; qc < 3; ++qc) {
      const double szv = tS[qc * 3 + c], dzv = tD[qc * 3 + c], pz = tP[qc * 2 + pc];
#pragma unroll 1
      for (int qb = 0; qb < 3; ++qb) {
        const double syv = tS[qb * 3 + b], dyv = tD[qb * 3 + b], py = tP[qb * 2 + pb];
        const double syz = syv * szv, dyz = dyv * szv, sdz = syv * dzv;
#pragma unroll
        for (int qa = 0; qa < 3; ++qa) {
          const int q = qa + 3 * (qb + 3 * qc);
          const double sxv = tS[qa * 3 + a], dxv = tD[qa * 3 + a];
          const double gx = dxv * syz, gy = sxv * dyz, gz = sxv * sdz, val = sxv * syz;
          const double *f = fl + q * 13;
#pragma unroll
          for (int comp = 0; comp < 3; ++comp) {
            rK[comp] = fma(gx, f[comp * 3], fma(gy, f[comp * 3 + 1], fma(gz, f[comp * 3 + 2], rK[comp])));
            rM[comp] = fma(val, f[10 + comp], rM[comp]);
          }
          if (t < 8) rP = fma(tP[qa * 2 + pa] * py * pz, f[9], rP);
        }
      }
    }
    // ---- distribute_local_to_global (constrained velocity rows are not written)
    for (int j = 0; j < prm.nout; ++j) {
      if (!con && prm.out_u[j]) {
#pragma unroll
        for (int comp = 0; comp < 3; ++comp)
          atomicAdd(prm.out_u[j] + comp * prm.Nu + gu, prm.wKu[j] * rK[comp] + prm.wM[j] * rM[comp]);
      }
      if (t < 8 && prm.out_p[j]) atomicAdd(prm.out_p[j] + gp, prm.wKp[j] * rP);
    }
  }
}

} // namespace

struct stfem_stokes_ctx {
  int device = 0;
  int nc[3] = {0, 0, 0};
  int ndu[3] = {0, 0, 0}, ndp[3] = {0, 0, 0};
  long long Nu = 0, Np = 0;
  int dmask = 0;
  double nu = 1.0;
  double *d_vertices = nullptr;
  StokesParams base;
};

static thread_local char g_stokes_err[256] = "";
#define STOKES_TRY(call)                                                   \
  do {                                                                     \
    hipError_t e_ = (call);                                                \
    if (e_ != hipSuccess) {                                                \
      snprintf(g_stokes_err, sizeof(g_stokes_err), "%s: %s", #call, hipGetErrorString(e_)); \
      return STFEM_ERR_HIP;                                                \
    }                                                                      \
  } while (0)

extern "C" {

const char *stfem_stokes_last_hip_error(void) { return g_stokes_err; }

int stfem_stokes_create(const stfem_mesh_desc *mesh, int velocity_degree, double viscosity, stfem_stokes_ctx **out)
{
  if (!mesh || !out) return STFEM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (velocity_degree != 2) return STFEM_ERR_UNSUPPORTED; // Q2/Q1 (BASELINE configs[4]) only
  for (int d = 0; d < 3; ++d)
    if (mesh->ncell[d] < 1) return STFEM_ERR_INVALID_ARGUMENT;
  int ndev = 0;
  {
    const hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
      snprintf(g_stokes_err, sizeof(g_stokes_err), "hipGetDeviceCount: %s (%d devices)", hipGetErrorString(e), ndev);
      return STFEM_ERR_NO_DEVICE;
    }
  }
  if (mesh->device < 0 || mesh->device >= ndev) return STFEM_ERR_INVALID_ARGUMENT;
  STOKES_TRY(hipSetDevice(mesh->device));
  stfem_stokes_ctx *c = new (std::nothrow) stfem_stokes_ctx;
  if (!c) return STFEM_ERR_OUT_OF_MEMORY;
  c->device = mesh->device;
  c->dmask = mesh->dirichlet_mask;
  c->nu = viscosity;
  for (int d = 0; d < 3; ++d) {
    c->nc[d] = mesh->ncell[d];
    c->ndu[d] = 2 * mesh->ncell[d] + 1;
    c->ndp[d] = mesh->ncell[d] + 1;
  }
  c->Nu = (long long)c->ndu[0] * c->ndu[1] * c->ndu[2];
  c->Np = (long long)c->ndp[0] * c->ndp[1] * c->ndp[2];
  const size_t nv = size_t(c->nc[0] + 1) * (c->nc[1] + 1) * (c->nc[2] + 1);
  std::vector<double> v(nv * 3);
  if (mesh->vertices) {
    std::memcpy(v.data(), mesh->vertices, nv * 3 * sizeof(double));
  } else {
    size_t o = 0;
    for (int k = 0; k <= c->nc[2]; ++k)
      for (int j = 0; j <= c->nc[1]; ++j)
        for (int i = 0; i <= c->nc[0]; ++i, ++o) {
          v[3 * o] = mesh->lower[0] + (mesh->upper[0] - mesh->lower[0]) * i / c->nc[0];
          v[3 * o + 1] = mesh->lower[1] + (mesh->upper[1] - mesh->lower[1]) * j / c->nc[1];
          v[3 * o + 2] = mesh->lower[2] + (mesh->upper[2] - mesh->lower[2]) * k / c->nc[2];
        }
  }
  if (hipMalloc(&c->d_vertices, nv * 3 * sizeof(double)) != hipSuccess) {
    delete c;
    return STFEM_ERR_OUT_OF_MEMORY;
  }
  if (hipMemcpy(c->d_vertices, v.data(), nv * 3 * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(c->d_vertices);
    delete c;
    return STFEM_ERR_HIP;
  }
  // 1D tables: FE_Q(2) and FE_Q(1) on Gauss-Lobatto nodes at the 3 Gauss points
  StokesParams &b = c->base;
  std::memset(&b, 0, sizeof(b));
  const stfem::ShapeTables tu = stfem::make_shape_tables(2), tp = stfem::make_shape_tables(1);
  std::vector<double> xq, wq;
  stfem::gauss_rule(3, xq, wq);
  stfem::Mat Sp, Gp;
  stfem::lagrange_tables(tp.nodes, xq, Sp, Gp);
  for (int i = 0; i < 9; ++i) { b.Su[i] = tu.S[i]; b.Du[i] = tu.D[i]; }
  for (int i = 0; i < 6; ++i) b.Sp[i] = Sp[i];
  for (int i = 0; i < 3; ++i) { b.xq[i] = xq[i]; b.wq[i] = wq[i]; }
  b.vertices = c->d_vertices;
  b.ncx = c->nc[0]; b.ncy = c->nc[1]; b.ncz = c->nc[2];
  for (int d = 0; d < 3; ++d) { b.ndu[d] = c->ndu[d]; b.ndp[d] = c->ndp[d]; }
  b.Nu = c->Nu; b.Np = c->Np;
  b.dmask = c->dmask;
  b.nu = c->nu;
  *out = c;
  return STFEM_OK;
}

void stfem_stokes_destroy(stfem_stokes_ctx *c)
{
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->d_vertices) (void)hipFree(c->d_vertices);
  delete c;
}

int64_t stfem_stokes_n_velocity_dofs(const stfem_stokes_ctx *c) { return c ? c->Nu : 0; }
int64_t stfem_stokes_n_pressure_dofs(const stfem_stokes_ctx *c) { return c ? c->Np : 0; }

static size_t stokes_len(const stfem_stokes_ctx *c, int variable) { return variable == 0 ? size_t(3 * c->Nu) : size_t(c->Np); }

int stfem_stokes_vector_create(stfem_stokes_ctx *c, int variable, double **device_out)
{
  if (!c || !device_out || variable < 0 || variable > 1) return STFEM_ERR_INVALID_ARGUMENT;
  *device_out = nullptr;
  STOKES_TRY(hipSetDevice(c->device));
  double *d = nullptr;
  if (hipMalloc(&d, stokes_len(c, variable) * sizeof(double)) != hipSuccess) return STFEM_ERR_OUT_OF_MEMORY;
  if (hipMemset(d, 0, stokes_len(c, variable) * sizeof(double)) != hipSuccess) {
    (void)hipFree(d);
    return STFEM_ERR_HIP;
  }
  *device_out = d;
  return STFEM_OK;
}

void stfem_stokes_vector_destroy(stfem_stokes_ctx *c, double *device_vec)
{
  if (!c || !device_vec) return;
  (void)hipSetDevice(c->device);
  (void)hipFree(device_vec);
}

int stfem_stokes_vector_upload(stfem_stokes_ctx *c, int variable, double *device_vec, const double *host)
{
  if (!c || !device_vec || !host || variable < 0 || variable > 1) return STFEM_ERR_INVALID_ARGUMENT;
  STOKES_TRY(hipSetDevice(c->device));
  STOKES_TRY(hipMemcpy(device_vec, host, stokes_len(c, variable) * sizeof(double), hipMemcpyHostToDevice));
  return STFEM_OK;
}

int stfem_stokes_vector_download(stfem_stokes_ctx *c, int variable, const double *device_vec, double *host)
{
  if (!c || !device_vec || !host || variable < 0 || variable > 1) return STFEM_ERR_INVALID_ARGUMENT;
  STOKES_TRY(hipSetDevice(c->device));
  STOKES_TRY(hipMemcpy(host, device_vec, stokes_len(c, variable) * sizeof(double), hipMemcpyDeviceToHost));
  return STFEM_OK;
}

static int stokes_launch(stfem_stokes_ctx *c, StokesParams &prm, hipStream_t st)
{
  const long long ncells = (long long)c->nc[0] * c->nc[1] * c->nc[2];
  (void)hipGetLastError();
  hipLaunchKernelGGL(stokes_cell_kernel, dim3((unsigned)((ncells + 7) / 8)), dim3(256), 0, st, prm);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_stokes_err, sizeof(g_stokes_err), "stokes_cell_kernel: %s", hipGetErrorString(e));
    return STFEM_ERR_HIP;
  }
  return STFEM_OK;
}

int stfem_stokes_vmult(stfem_stokes_ctx *c, double *dst_u, double *dst_p, const double *src_u,
                       const double *src_p, void *stream)
{
  if (!c || !dst_u || !dst_p || !src_u || !src_p) return STFEM_ERR_INVALID_ARGUMENT;
  if (dst_u == src_u || dst_p == src_p) return STFEM_ERR_ALIAS;
  STOKES_TRY(hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  STOKES_TRY(hipMemsetAsync(dst_u, 0, sizeof(double) * 3 * c->Nu, st));
  STOKES_TRY(hipMemsetAsync(dst_p, 0, sizeof(double) * c->Np, st));
  StokesParams prm = c->base;
  prm.u = src_u; prm.p = src_p;
  prm.nout = 1;
  prm.out_u[0] = dst_u; prm.out_p[0] = dst_p;
  prm.wKu[0] = 1.0; prm.wKp[0] = 1.0; prm.wM[0] = 0.0;
  return stokes_launch(c, prm, st);
}

int stfem_stokes_mass_vmult(stfem_stokes_ctx *c, double *dst_u, const double *src_u, void *stream)
{
  if (!c || !dst_u || !src_u) return STFEM_ERR_INVALID_ARGUMENT;
  if (dst_u == src_u) return STFEM_ERR_ALIAS;
  STOKES_TRY(hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  STOKES_TRY(hipMemsetAsync(dst_u, 0, sizeof(double) * 3 * c->Nu, st));
  StokesParams prm = c->base;
  prm.u = src_u; prm.p = nullptr;
  prm.nout = 1;
  prm.out_u[0] = dst_u; prm.out_p[0] = nullptr;
  prm.wKu[0] = 0.0; prm.wKp[0] = 0.0; prm.wM[0] = 1.0;
  return stokes_launch(c, prm, st);
}

int stfem_stokes_st_vmult(stfem_stokes_ctx *c, int n_timesteps_at_once, int n_timedofs, int variable_major,
                          const double *Alpha, const double *Beta, double *const *dst_blocks,
                          const double *const *src_blocks, void *stream)
{
  if (!c || !Alpha || !Beta || !dst_blocks || !src_blocks || n_timesteps_at_once < 1 || n_timedofs < 1)
    return STFEM_ERR_INVALID_ARGUMENT;
  const int nt = n_timedofs, ns = n_timesteps_at_once, nb = 2 * nt * ns;
  auto index = [&](int it, int v, int d) { // BlockSlice::index, fe_time.h:956-967
    return variable_major ? it * (2 * nt) + v * nt + d : it * (2 * nt) + d * 2 + v;
  };
  for (int j = 0; j < nb; ++j) {
    if (!dst_blocks[j] || !src_blocks[j]) return STFEM_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < nb; ++i)
      if (dst_blocks[j] == src_blocks[i]) return STFEM_ERR_ALIAS;
  }
  STOKES_TRY(hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  for (int it = 0; it < ns; ++it) // dst = 0.0 (operators.h:833)
    for (int d = 0; d < nt; ++d) {
      STOKES_TRY(hipMemsetAsync(dst_blocks[index(it, 0, d)], 0, sizeof(double) * 3 * c->Nu, st));
      STOKES_TRY(hipMemsetAsync(dst_blocks[index(it, 1, d)], 0, sizeof(double) * c->Np, st));
    }
  const double eps10 = 10 * std::numeric_limits<double>::epsilon(); // internal::scatter, operators.h:106
  for (int it = 0; it < ns; ++it)
    for (int id = 0; id < nt; ++id) {
      const int i = index(it, 0, id); // the velocity column drives all scatters (operators.h:851-862)
      StokesParams prm = c->base;
      prm.u = src_blocks[index(it, 0, id)];
      prm.p = src_blocks[index(it, 1, id)];
      prm.nout = 0;
      auto flush = [&]() -> int {
        if (prm.nout == 0) return STFEM_OK;
        const int rc = stokes_launch(c, prm, st);
        prm.nout = 0;
        return rc;
      };
      for (int jt = 0; jt < ns; ++jt)
        for (int jd = 0; jd < nt; ++jd) {
          const int ju = index(jt, 0, jd), jp = index(jt, 1, jd);
          const double aU = Alpha[size_t(ju) * nb + i], aP = Alpha[size_t(jp) * nb + i], bU = Beta[size_t(ju) * nb + i];
          const bool useU = std::abs(aU) > eps10, useP = std::abs(aP) > eps10, useM = std::abs(bU) > eps10;
          if (!useU && !useP && !useM) continue;
          const int o = prm.nout++;
          prm.out_u[o] = (useU || useM) ? dst_blocks[ju] : nullptr;
          prm.out_p[o] = useP ? dst_blocks[jp] : nullptr;
          prm.wKu[o] = useU ? aU : 0.0;
          prm.wKp[o] = useP ? aP : 0.0;
          prm.wM[o] = useM ? bU : 0.0;
          if (prm.nout == MAXOUT) {
            const int rc = flush();
            if (rc != STFEM_OK) return rc;
          }
        }
      const int rc = flush();
      if (rc != STFEM_OK) return rc;
    }
  return STFEM_OK;
}

// SystemMatrixStokes::vmult_slice_add (operators.h:748-781): the n x 1 case used for the right-hand
// side: src = one (velocity, pressure) pair, dst[index(it,v,id)] += Gamma(index(it,v,id), 0) * (K_S src)_v
// and dst[index(it,0,id)] += Zeta(index(it,0,id), 0) * M u.  dst is NOT zeroed.
int stfem_stokes_st_vmult_slice_add(stfem_stokes_ctx *c, int n_timesteps_at_once, int n_timedofs, int variable_major,
                                    const double *Gamma, const double *Zeta, double *const *dst_blocks,
                                    const double *src_u, const double *src_p, void *stream)
{
  if (!c || !Gamma || !Zeta || !dst_blocks || !src_u || !src_p || n_timesteps_at_once < 1 || n_timedofs < 1)
    return STFEM_ERR_INVALID_ARGUMENT;
  const int nt = n_timedofs, ns = n_timesteps_at_once, nb = 2 * nt * ns;
  auto index = [&](int it, int v, int d) { return variable_major ? it * (2 * nt) + v * nt + d : it * (2 * nt) + d * 2 + v; };
  for (int j = 0; j < nb; ++j) {
    if (!dst_blocks[j]) return STFEM_ERR_INVALID_ARGUMENT;
    if (dst_blocks[j] == src_u || dst_blocks[j] == src_p) return STFEM_ERR_ALIAS;
  }
  STOKES_TRY(hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const double eps10 = 10 * std::numeric_limits<double>::epsilon();
  StokesParams prm = c->base;
  prm.u = src_u;
  prm.p = src_p;
  prm.nout = 0;
  for (int it = 0; it < ns; ++it)
    for (int id = 0; id < nt; ++id) {
      const int ju = index(it, 0, id), jp = index(it, 1, id);
      const double aU = Gamma[ju], aP = Gamma[jp], bU = Zeta[ju];
      const bool useU = std::abs(aU) > eps10, useP = std::abs(aP) > eps10, useM = std::abs(bU) > eps10;
      if (!useU && !useP && !useM) continue;
      const int o = prm.nout++;
      prm.out_u[o] = (useU || useM) ? dst_blocks[ju] : nullptr;
      prm.out_p[o] = useP ? dst_blocks[jp] : nullptr;
      prm.wKu[o] = useU ? aU : 0.0;
      prm.wKp[o] = useP ? aP : 0.0;
      prm.wM[o] = useM ? bU : 0.0;
      if (prm.nout == MAXOUT) {
        const int rc = stokes_launch(c, prm, st);
        if (rc != STFEM_OK) return rc;
        prm.nout = 0;
      }
    }
  return prm.nout ? stokes_launch(c, prm, st) : STFEM_OK;
}

} // extern "C"
