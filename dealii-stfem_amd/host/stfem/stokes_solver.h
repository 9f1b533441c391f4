// The solver side of the Stokes slab problem (tests/tp_03stokes.cc:536-560, 840-1090) over the C-ABI: the two-variable block vector
// with the vector arithmetic of the Krylov solver, the relaxation smoother around the two-variable Vanka smoother, and the first-order
// time integrator of include/time_integrators.h:30-336 with a velocity force, for FE_Q(2)^3 x FE_Q(1) (BASELINE configs[4]).
// Control flow only: every vector operation, integral and operator application is a call into libstfem_hip.so.
#pragma once
#include "stokes.h"
#include "time_integrators.h"

namespace stfem {

// The scalar spaces behind the two variables: a velocity component is a FE_Q(2) function (with the velocity's strong constraints),
// the pressure a FE_Q(1) function without constraints.  Their contexts serve the vector arithmetic, the load vectors, the
// interpolation and the error norms of the Stokes vectors (VectorTools::* per variable in the reference).
struct StokesSpaces {
  std::shared_ptr<Context> q2, q1;
  explicit StokesSpaces(const Mesh &mesh)
  {
    MatrixFreeOperator<3, 1, double> a(mesh, 2, 0.0, 1.0);
    q2 = a.context();
    Mesh m1 = mesh;
    m1.dirichlet_mask = 0;
    MatrixFreeOperator<3, 1, double> b(m1, 1, 1.0, 0.0);
    q1 = b.context();
  }
};

// BlockVectorT over a two-variable BlockSlice (LinearAlgebra::distributed::BlockVector with velocity and pressure blocks)
class StokesBlockVector {
public:
  StokesBlockVector() = default;
  void reinit(const std::shared_ptr<StokesSpaces> &spaces, stfem_stokes_ctx *stokes, const BlockSlice &slice)
  {
    spaces_ = spaces; stokes_ = stokes; slice_ = std::make_shared<BlockSlice>(slice);
    blocks_.clear();
    views_.clear();
    const size_t nu = size_t(stfem_stokes_n_velocity_dofs(stokes));
    for (unsigned i = 0; i < slice.n_blocks(); ++i) {
      const int var = int(slice.decompose(i)[1]);
      blocks_.emplace_back(stokes, var);
      BlockVectorT<double> v;
      if (var == 0) {
        void *ptrs[3] = {blocks_.back().data(), blocks_.back().data() + nu, blocks_.back().data() + 2 * nu};
        v.wrap(spaces->q2, ptrs, 3);
      } else {
        void *ptrs[1] = {blocks_.back().data()};
        v.wrap(spaces->q1, ptrs, 1);
      }
      views_.push_back(std::move(v));
    }
  }
  bool empty() const { return blocks_.empty(); }
  unsigned n_blocks() const { return unsigned(blocks_.size()); }
  std::vector<StokesVector> &blocks() { return blocks_; }
  const std::vector<StokesVector> &blocks() const { return blocks_; }
  // block i as a block vector of its scalar space (velocity: the three components)
  BlockVectorT<double> &view(unsigned i) { return views_.at(i); }
  const BlockVectorT<double> &view(unsigned i) const { return views_.at(i); }
  const std::shared_ptr<StokesSpaces> &spaces() const { return spaces_; }
  stfem_stokes_ctx *stokes() const { return stokes_; }
  const BlockSlice &slice() const { return *slice_; }

private:
  std::shared_ptr<StokesSpaces> spaces_;
  stfem_stokes_ctx *stokes_ = nullptr;
  std::shared_ptr<BlockSlice> slice_;
  std::vector<StokesVector> blocks_;
  std::vector<BlockVectorT<double>> views_;
};

inline void reinit_like(StokesBlockVector &v, const StokesBlockVector &x)
{
  if (v.empty()) v.reinit(x.spaces(), x.stokes(), x.slice());
}
inline void axpby(double a, const StokesBlockVector &x, double b, StokesBlockVector &y, void *stream = nullptr)
{
  for (unsigned i = 0; i < y.n_blocks(); ++i) axpby(a, x.view(i), b, y.view(i), stream);
}
inline void set_zero(StokesBlockVector &v, void *stream = nullptr)
{
  for (unsigned i = 0; i < v.n_blocks(); ++i) set_zero(v.view(i), stream);
}
inline double dot(const StokesBlockVector &a, const StokesBlockVector &b)
{
  double s = 0.0;
  for (unsigned i = 0; i < a.n_blocks(); ++i) s += dot(a.view(i), b.view(i));
  return s;
}
inline double norm(const StokesBlockVector &x) { return std::sqrt(dot(x, x)); }
// the Gram-Schmidt step of the Krylov solver (modified scheme: one inner product and one update per basis vector)
inline double orthogonalize(const std::vector<StokesBlockVector> &vs, unsigned k, StokesBlockVector &w, double *h)
{
  for (unsigned i = 0; i < k; ++i) {
    h[i] = dot(w, vs[i]);
    axpby(-h[i], vs[i], 1.0, w);
  }
  return norm(w);
}

// SystemMatrixStokes on StokesBlockVector (the operator interface the solver consumes)
template <int dim, typename Number> class StokesSystem {
public:
  StokesSystem(const SystemMatrixStokes<dim, Number> &A, const std::shared_ptr<StokesSpaces> &spaces, stfem_stokes_ctx *stokes, const BlockSlice &slice)
    : A(A), spaces(spaces), stokes(stokes), slice(slice)
  {}
  void initialize_dof_vector(StokesBlockVector &v) const { v.reinit(spaces, stokes, slice); }
  void vmult(StokesBlockVector &dst, const StokesBlockVector &src, void *stream = nullptr) const { A.vmult(dst.blocks(), src.blocks(), stream); }

private:
  const SystemMatrixStokes<dim, Number> &A;
  std::shared_ptr<StokesSpaces> spaces;
  stfem_stokes_ctx *stokes;
  BlockSlice slice;
};

// PreconditionRelaxation around the two-variable Vanka smoother (stmg.h:1199-1238): n sweeps of x <- x + omega P^-1 (b - A x) from 0
template <typename System> class PreconditionRelaxationStokes {
public:
  PreconditionRelaxationStokes(const System &A, const PreconditionVankaStokes<double> &P, double omega, unsigned n_iterations)
    : A(A), P(P), omega(omega), n_iterations(n_iterations)
  {}
  void vmult(StokesBlockVector &dst, const StokesBlockVector &src, void *stream = nullptr) const
  {
    P.step(dst.blocks(), omega, false, src.blocks(), stream);
    for (unsigned it = 1; it < n_iterations; ++it) {
      reinit_like(res, src);
      A.vmult(res, dst, stream);
      axpby(1.0, src, -1.0, res, stream);
      P.step(dst.blocks(), omega, true, res.blocks(), stream);
    }
  }

private:
  const System &A;
  const PreconditionVankaStokes<double> &P;
  double omega;
  unsigned n_iterations;
  mutable StokesBlockVector res;
};

// A vector function of (x, t) at a list of points: out[c][i] = f_c(points[3 i .. 3 i + 2], t)
using VectorPointFunction = std::function<void(double time, const std::vector<double> &points, std::array<std::vector<double>, 3> &out)>;

// include/time_integrators.h:30-336 for the two-variable system: rhs = rhs_matrix (prev_u, prev_p) + time quadrature of the velocity
// force (assemble_force per variable, 73-111: the pressure load is zero), FGMRES on the slab system, the pressure of every time
// dof shifted to zero mean afterwards (tests/tp_03stokes.cc:1047-1062).  Alpha_1 / Gamma_1: the ONE-step scalar temporal matrices.
template <int dim, typename System, typename Preconditioner> class TimeIntegratorStokes {
public:
  TimeIntegratorStokes(TimeStepType type, unsigned time_degree, const FullMatrix<double> &Alpha_1, const FullMatrix<double> &Gamma_1,
                       double gmres_tolerance, const System &matrix, const Preconditioner &preconditioner,
                       const SystemMatrixStokes<dim, double> &rhs_matrix, const VectorPointFunction &force, bool zero_mean_pressure,
                       double abstol = 1e-12, unsigned max_steps = 400)
    : type(type), time_degree(time_degree), quad_time(time_points(type, time_degree)), Alpha(Alpha_1), Gamma(Gamma_1),
      solver(max_steps, abstol, gmres_tolerance, 200), preconditioner(preconditioner), matrix(matrix), rhs_matrix(rhs_matrix), force(force),
      nt_dofs(type == TimeStepType::DG ? time_degree + 1 : time_degree), zero_mean(zero_mean_pressure)
  {
    if (const char *e = std::getenv("STFEM_FGMRES_VERBOSE")) solver.verbose = unsigned(std::atoi(e));
  }

  // x, rhs: the slab's blocks; prev: one (velocity, pressure) pair (BlockSlice(1, 2, 1))
  void solve(StokesBlockVector &x, const StokesBlockVector &prev, StokesBlockVector &rhs, double time, double time_step)
  {
    TraceRange scope("step");
    const StokesSpaces &sp = *x.spaces();
    const BlockSlice &slice = x.slice();
    set_zero(rhs);
    rhs_matrix.vmult_slice_add(rhs.blocks(), prev.blocks());
    // assemble_force: Alpha is diagonal (time quadrature = support points)
    if (qpoints.empty()) {
      qpoints.resize(size_t(stfem_n_cells(sp.q2->h)) * 27 * 3);
      check(stfem_quadrature_points(sp.q2->h, 3, qpoints.data()), "stfem_quadrature_points");
      load.reinit(sp.q2, 3);
    }
    std::array<std::vector<double>, 3> fq;
    for (unsigned j = 0; j < quad_time.size(); ++j) {
      force(time + time_step * quad_time[j], qpoints, fq);
      for (int c = 0; c < 3; ++c) check(stfem_integrate_rhs(sp.q2->h, 3, fq[c].data(), load.handle(), c, nullptr), "stfem_integrate_rhs");
      auto add = [&](unsigned timedof, double w) { axpby(w, load, 1.0, rhs.view(slice.index(0, 0, timedof))); };
      if (type == TimeStepType::DG) add(j, Alpha(j, j));
      else if (j == 0)
        for (unsigned i = 0; i < nt_dofs; ++i) add(i, -Gamma(i, 0));
      else add(j - 1, Alpha(j - 1, j - 1));
    }
    for (unsigned i = 0; i < x.n_blocks(); ++i) // extrapolate (time_integrators.h:184-194): every time dof starts from the previous solution
      axpby(1.0, prev.view(slice.decompose(i)[1]), 0.0, x.view(i));
    solver.solve(matrix, x, rhs, preconditioner);
    if (zero_mean) {
      if (!weights.handle()) { // 1^T M_p: the load vector of the constant one; |Omega| = 1^T M_p 1
        weights.reinit(sp.q1, 1);
        ones.reinit(sp.q1, 1);
        const size_t nq = size_t(stfem_n_cells(sp.q1->h)) * 8;
        std::vector<double> one(nq, 1.0);
        check(stfem_integrate_rhs(sp.q1->h, 2, one.data(), weights.handle(), 0, nullptr), "stfem_integrate_rhs");
        std::vector<std::vector<double>> h1(1, std::vector<double>(ones.block_size(), 1.0));
        ones.copy_from_host(h1);
        volume = dot(weights, ones);
      }
      for (unsigned a = 0; a < nt_dofs; ++a) {
        BlockVectorT<double> &p = x.view(slice.index(0, 1, a));
        axpby(-dot(weights, p) / volume, ones, 1.0, p);
      }
    }
  }
  unsigned last_step() const { return solver.last_step(); }

private:
  TimeStepType type;
  unsigned time_degree;
  std::vector<double> quad_time;
  const FullMatrix<double> &Alpha, &Gamma;
  SolverFGMRES<double, StokesBlockVector> solver;
  const Preconditioner &preconditioner;
  const System &matrix;
  const SystemMatrixStokes<dim, double> &rhs_matrix;
  VectorPointFunction force;
  unsigned nt_dofs;
  bool zero_mean;
  std::vector<double> qpoints;
  BlockVectorT<double> load, weights, ones;
  double volume = 1.0;
};

} // namespace stfem
