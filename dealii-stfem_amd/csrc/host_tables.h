// Host-side numerics of the product library: 1D rules, shape tables in even-odd form,
// temporal matrices (fe_time.h of the reference), mesh and coefficient helpers.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace stfem {

// ---- 1D rules on [0,1] (deal.II QGauss, QGaussLobatto, QGaussRadau(right)) ----
void gauss_rule(int n, std::vector<double> &x, std::vector<double> &w);
std::vector<double> lobatto_points(int n);
std::vector<double> radau_right_points(int n);

using Mat = std::vector<double>; // row-major

// Lagrange basis on `nodes` evaluated at `x`: V[q*n+a] = l_a(x_q), G[q*n+a] = l_a'(x_q)
void lagrange_tables(const std::vector<double> &nodes, const std::vector<double> &x, Mat &V, Mat &G);

// Even-odd packing of an n x n matrix X with X[q][a] = sign * X[n-1-q][n-1-a].
// Layout (h = n/2): ee[q*h+i] | oo[q*h+i] | em[q] | mm[i] | mmm     (last three only for odd n)
constexpr int eo_size(int n) { return 2 * (n / 2) * (n / 2) + ((n & 1) ? 2 * (n / 2) + 1 : 0); }
constexpr int EO_MAX = 32; // >= eo_size(7)
void eo_pack(int n, const Mat &X, double *out);

// Per-degree kernel tables.  Quadrature weights are folded into the tables:
//   Si  = diag(sqrt w) S            (nodes -> Gauss points, "interpolate")
//   Dc  = diag(sqrt w) Dcol diag(1/sqrt w)   (collocation derivative at Gauss points)
// so that M_cell = vol * Si^T Si and K_cell,d = vol/h_d^2 * Si^T Dc^T Dc Si per direction.
struct ShapeTables {
  int n; // p+1 == n_q
  Mat S, D;      // plain S[q][a] = l_a(x_q), D[q][a] = l_a'(x_q)   (nodal -> Gauss)
  Mat Dcol;      // plain collocation derivative on the Gauss points
  std::vector<double> xq, wq, nodes;
  double eo_Si[EO_MAX], eo_SiT[EO_MAX], eo_Dc[EO_MAX], eo_DcT[EO_MAX];
  double eo_L[EO_MAX]; // L = Dc^T Dc: 1D weighted collocation Laplacian (symmetric type)
  double eo_S[EO_MAX], eo_ST[EO_MAX], eo_Dq[EO_MAX], eo_DqT[EO_MAX]; // unweighted S, Dcol
  // Simultaneous diagonalisation of the 1D nodal mass and stiffness matrices of the reference
  // cell, M1 = S^T diag(w) S and K1 = D^T diag(w) D:  W M1^-1 W^T = I,  K1 = W^T diag(lam) W
  // (W = V^-1 for the M1-orthonormal generalised eigenvectors V).  Both matrices are
  // persymmetric, so every mode is even or odd: rows 0..ne-1 of W are even (ne = ceil(n/2)),
  // rows ne..n-1 odd.  fd_W packs only the independent half:
  //   We[m*ne + i], i < ne   (coefficient of u_i + u_{n-1-i}, and of u_{n/2} for the middle)
  //   Wo[m*h + i],  i < h    (coefficient of u_i - u_{n-1-i}),   h = n/2, stored after We
  double fd_W[EO_MAX], fd_lam[8];
};
ShapeTables make_shape_tables(int p);

// ---- temporal matrices ----
int time_nb(int type, int r, int nsteps);
int fe_time_weights(int type, int r, double tau, int nsteps, Mat &Alpha, Mat &Beta, Mat &Gamma,
                    Mat &Zeta);
int fe_time_weights_wave(int type, int r, double tau, int nsteps, Mat &A_lhs, Mat &B_lhs,
                         Mat &rhs_uK, Mat &rhs_uM, Mat &rhs_vM);

// time-multigrid transfer matrices (fe_time.h:749-898), row-major m x n; return 0 or -1 (bad arguments)
int time_prolongation(int type, int r, int nsteps, Mat &out, int &m, int &n);
int time_restriction(int type, int r, int nsteps, Mat &out, int &m, int &n);
int time_projection(int type, int r_src, int r_dst, int nsteps, Mat &out, int &m, int &n);

// ---- level schedule of the space-time multigrid (fe_time.cc:17-150); levels are 't' (tau), 'k', 'h', 'p' like MGType ----
// get_poly_mg_sequence: sequence_type 0 = bisect, 1 = decrease_by_one, 2 = go_to_one; coarsest first
std::vector<int> poly_mg_sequence(int k_max, int k_min, int sequence_type);
// get_mg_sequence: n_k / n_p = lengths of the temporal / spatial degree sequences; coarsest transfer first
std::string mg_sequence(int n_sp_lvl, int n_k, int n_p, int n_timesteps_at_once, int n_timesteps_at_once_min, char lower_lvl,
                        int coarsening_type, bool time_before_space, bool use_p_multigrid_space, bool zip_from_back);
// get_precondition_stmg_types: one smoother id per level (0 = identity)
std::vector<int> precondition_stmg_types(const std::string &mg_type_level, int coarsening_type, bool time_before_space, int smoother);

// ---- mesh / coefficient ----
void mesh_vertices(const int32_t gn[3], const double lo[3], const double up[3], double distort,
                   uint64_t seed, int32_t z0, int32_t z1, double *out);
void coefficient_per_cell(const int32_t nc[3], const double *vertices, double c1, double c2,
                          double c3, double distort, const int32_t sub[3], const double lo[3],
                          const double up[3], double *out);

} // namespace stfem
