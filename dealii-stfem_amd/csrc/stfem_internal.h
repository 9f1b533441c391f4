// Internal layout of the opaque handles of include/stfem.h, shared by the translation units of the
// library (stfem_capi.hip, stfem_vanka.hip).  Not part of the boundary.
#pragma once
#include "../../include/stfem.h"

#include "host_tables.h"

#include <cstdint>
#include <vector>

struct stfem_ctx {
  int p = 0, device = 0, n_cu = 0;
  int nc[3] = {0, 0, 0}, nd[3] = {0, 0, 0};
  int64_t ndofs = 0, ncells = 0;
  int dmask = 0;
  bool cartesian = false;
  double lower[3] = {0, 0, 0}, h[3] = {1, 1, 1};
  stfem::ShapeTables tab;
  std::vector<double> vertices; // host copy (general meshes)
  int prec = 0;                // 0 = fp64, 1 = fp32 (element type of vectors, coefficients, metric)
  size_t es = sizeof(double);  // element size
  void *d_coef[2] = {nullptr, nullptr}; // [0] mass, [1] laplace
  int coef_layout[2] = {0, 0};
  double *d_scratch = nullptr; // reductions (always double)
  const char *last_kernel = "";
  // tile variant: halo slabs (grown on demand)
  void *d_halo = nullptr;
  size_t halo_doubles = 0; // elements
  int variant = 0; // 0 = pencil (default; tile where the pencil kernel has no instantiation), 1 = atomic, 2 = tile
  // tuning / experiment switches, read once at context creation (STFEM_* environment variables)
  int env_tile_lz = 0, env_exp = 0, env_stagger = 0, env_stagger_div = 256, env_pencil_ty = 0, env_pencil_lz = 0;
  const char *env_timeline = nullptr;
  int *d_work = nullptr;           // pencil variant: tile counters
  long long *d_timeline = nullptr; // diagnostic builds: phase timestamps of the last apply
  size_t tl_n = 0;
  // general-geometry path: device copies of vertices and the 1D rule, metric terms per (cell, q)
  double *d_vertices = nullptr, *d_rule = nullptr;
  void *d_metric = nullptr;
  bool metric_valid = false;
  int metric_flags = -1; // which coefficients are baked into d_metric (bit0 laplace, bit1 mass)
  // csrc/stfem_stokes.hip: the next store-mode sweep of three FE_Q(2) blocks adds - grad_scale B^T p (SweepParams::gp); set and
  // cleared around one stfem_st_vmult by stfem_internal_set_gradient, grad_applied tells whether a launch took it
  const double *grad_p = nullptr;
  double grad_w[3][2][3][2];
  double grad_scale = 0.0;
  bool grad_applied = false;
};
int stfem_internal_set_gradient(stfem_ctx *c, const double *p, const double (*w)[2][3][2], double scale);

struct stfem_vec {
  stfem_ctx *ctx = nullptr;
  int nb = 0;
  bool owns = false;
  std::vector<void *> blk; // device arrays of the context's element type
};


// stfem_capi.hip: (re)builds the per-quadrature-point metric records [cell][qz][qy][qx][8] =
// (Gxx,Gxy,Gxz,Gyy,Gyz,Gzz,Mq,pad) with the coefficient tables in force; element type = the context's Number
int stfem_internal_metric(stfem_ctx *c, const void **metric, void *stream);

// stfem_stokes.hip: what stfem_stokes_vanka.hip needs to know about a Stokes context
struct stfem_stokes_desc {
  int device, cart, pspace, dmask, weak_mask, outflow_mask;
  int nc[3], ndu[3], ndp[3];
  long long Nu, Np;
  double lower[3], upper[3], nu, penalty1, penalty2;
};
int stfem_stokes_internal_desc(const stfem_stokes_ctx *c, stfem_stokes_desc *out);
