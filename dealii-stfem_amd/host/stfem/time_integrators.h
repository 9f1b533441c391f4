// Host-side mirror of the slab driver: the reference's TimeIntegratorFO
// (include/time_integrators.h:30-336), the pieces of deal.II it drives (SolverFGMRES with a ReductionControl,
// PreconditionRelaxation around the Vanka smoother) and ErrorCalculator (include/exact_solution.h:503-649),
// on top of the C-ABI.  Everything heavy - operator, smoother, load vectors, error norms, vector arithmetic -
// runs on the device; this file is the control flow.
#pragma once
#include "operators.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

namespace stfem {

// ---- vector arithmetic of the solver (LinearAlgebra::distributed::BlockVector::add / sadd / equ / l2_norm)
template <typename Number> void axpby(double a, const BlockVectorT<Number> &x, double b, BlockVectorT<Number> &y, void *stream = nullptr)
{
  check(stfem_vector_axpby(x.context()->h, a, x.handle(), b, y.handle(), stream), "stfem_vector_axpby");
}
// `v = 0.` (a memset: whatever v held, NaN included, is gone)
template <typename Number> void set_zero(BlockVectorT<Number> &v, void *stream = nullptr)
{
  check(stfem_vector_set_zero(v.context()->h, v.handle(), stream), "stfem_vector_set_zero");
}
template <typename Number> double norm(const BlockVectorT<Number> &x) { return std::sqrt(dot(x, x)); }

// The Gram-Schmidt step of the Krylov solvers: w is orthogonalised against v_0 .. v_{k-1}, h[i] receives the
// coefficients, the return value is ||w|| afterwards.  One rank: the classical scheme on the device with a second pass
// only when the first one cancelled most of w (||w|| fell below a tenth: the loss of orthogonality of one classical pass
// is eps ||w||_before / ||w||_after; deal.II's SolverGMRES re-orthogonalises on such a test too) - a handful of launches
// and one or two read-backs whatever k is.  Partitioned vectors: the modified scheme with one reducing inner product per
// vector (stfem_dot_global), as before.
template <typename Number> double orthogonalize(const std::vector<BlockVectorT<Number>> &vs, unsigned k, BlockVectorT<Number> &w, double *h)
{
  const Context &c = *w.context();
  if (c.comm || w.n_blocks() > 8 || k > 240) {
    for (unsigned i = 0; i < k; ++i) {
      h[i] = dot(w, vs[i]);
      axpby(-h[i], vs[i], 1.0, w);
    }
    return norm(w);
  }
  std::vector<const stfem_vec *> handles(k);
  for (unsigned i = 0; i < k; ++i) handles[i] = vs[i].handle();
  double before = 0.0, n2 = 0.0;
  check(stfem_orthogonalize(c.h, int(k), handles.data(), w.handle(), 0, h, &before, &n2, nullptr), "stfem_orthogonalize");
  if (!(n2 > 0.01 * before)) {
    std::vector<double> h2(k);
    check(stfem_orthogonalize(c.h, int(k), handles.data(), w.handle(), 0, h2.data(), nullptr, &n2, nullptr), "stfem_orthogonalize");
    for (unsigned i = 0; i < k; ++i) h[i] += h2[i];
  }
  return std::sqrt(std::max(n2, 0.0));
}

// One block of a block vector as a one-block vector of its own (a view: nothing is copied)
template <typename Number> BlockVectorT<Number> block_view(const BlockVectorT<Number> &v, unsigned b)
{
  BlockVectorT<Number> out;
  void *ptr = stfem_vector_block(v.handle(), int(b));
  out.wrap(v.context(), &ptr, 1);
  return out;
}

// Blocks [first, first + count) of a block vector as a vector of their own (a view)
template <typename Number> BlockVectorT<Number> block_range(const BlockVectorT<Number> &v, unsigned first, unsigned count)
{
  std::vector<void *> ptrs(count);
  for (unsigned b = 0; b < count; ++b) ptrs[b] = stfem_vector_block(v.handle(), int(first + b));
  BlockVectorT<Number> out;
  out.wrap(v.context(), ptrs.data(), count);
  return out;
}
// tensorproduct_add (include/operators.h:211-250): c[offset + i] += sum_j A(i, j) b[offset + j]  (b a block vector),
// or += A(i, 0) b (b one spatial vector)
template <typename Number>
void tensorproduct_add(BlockVectorT<Number> &c, const FullMatrix<Number> &A, const BlockVectorT<Number> &b, unsigned block_offset = 0)
{
  std::vector<double> a(size_t(A.m()) * A.n());
  for (size_t i = 0; i < a.size(); ++i) a[i] = double(A.data()[i]);
  BlockVectorT<Number> cv = block_range(c, block_offset, A.m());
  if (b.n_blocks() == 1 && A.n() == 1) {
    check(stfem_tensorproduct_add(c.context()->h, int(A.m()), 1, a.data(), cv.handle(), b.handle(), nullptr), "stfem_tensorproduct_add");
  } else {
    BlockVectorT<Number> bv = block_range(b, block_offset, A.n());
    check(stfem_tensorproduct_add(c.context()->h, int(A.m()), int(A.n()), a.data(), cv.handle(), bv.handle(), nullptr), "stfem_tensorproduct_add");
  }
}

struct PreconditionIdentity {
  template <typename V> void vmult(V &dst, const V &src) const { axpby(1.0, src, 0.0, dst); }
};

// PreconditionRelaxation (deal.II) with the Vanka smoother as inner preconditioner, as the multigrid levels of the
// reference use it (stmg.h:1199-1238): n_iterations sweeps of x <- x + omega P^-1 (b - A x) from x = 0
template <typename Number, typename Operator> class PreconditionRelaxation {
public:
  PreconditionRelaxation(const Operator &A, const PreconditionVanka<Number> &P, double omega, unsigned n_iterations)
    : A(A), P(P), omega(omega), n_iterations(n_iterations)
  {}
  void vmult(BlockVectorT<Number> &dst, const BlockVectorT<Number> &src, void *stream = nullptr) const
  {
    if (!res.handle() && n_iterations > 1) A.initialize_dof_vector(res);
    P.step(dst, omega, false, src, stream); // dst = omega P^-1 src (scaling and update ride in the smoother's scatter)
    for (unsigned it = 1; it < n_iterations; ++it) {
      A.vmult(res, dst, stream);
      axpby(1.0, src, -1.0, res, stream); // res = src - A dst
      P.step(dst, omega, true, res, stream);
    }
  }

private:
  const Operator &A;
  const PreconditionVanka<Number> &P;
  double omega;
  unsigned n_iterations;
  mutable BlockVectorT<Number> res;
};

// deal.II SolverFGMRES with ReductionControl(max_steps, abs_tol, reduce) as the reference sets it up
// (time_integrators.h:57-60: 200 steps, 1e-12 absolute, gmres_tolerance relative, restart 100):
// right-preconditioned flexible GMRES, modified Gram-Schmidt, Givens rotations.
// VectorType: BlockVectorT<Number>, or any type with the free functions axpby, norm, orthogonalize and reinit_like
// (host/stfem/stokes_solver.h: the two-variable block vector of the Stokes systems).
template <typename Number> inline void reinit_like(BlockVectorT<Number> &v, const BlockVectorT<Number> &x)
{
  if (!v.handle()) v.reinit(x.context(), x.n_blocks());
}
template <typename Number, typename VectorType = BlockVectorT<Number>> class SolverFGMRES {
public:
  using V = VectorType;
  SolverFGMRES(unsigned max_steps, double abs_tol, double reduce, unsigned restart = 100)
    : max_steps(max_steps), abs_tol(abs_tol), reduce(reduce), restart(restart)
  {}
  unsigned last_step() const { return steps; }
  double last_value() const { return value; }
  unsigned verbose = 0; // print the residual every `verbose` steps

  template <typename Operator, typename Preconditioner>
  void solve(const Operator &A, V &x, const V &b, const Preconditioner &P)
  {
    steps = 0;
    V r;
    A.initialize_dof_vector(r);
    auto fresh = [&](V &v) { reinit_like(v, x); };
    double tol = abs_tol;
    bool first = true;
    while (true) {
      A.vmult(r, x);
      axpby(1.0, b, -1.0, r);
      double beta = norm(r);
      if (first) {
        tol = std::max(abs_tol, reduce * beta);
        first = false;
      }
      value = beta;
      if (beta <= tol) return;
      if (steps >= max_steps) throw std::runtime_error("SolverFGMRES: no convergence");
      const unsigned m = std::min(restart, max_steps - steps);
      if (vs.size() < m + 1) { vs.resize(m + 1); zs.resize(m); }
      std::vector<double> H(size_t(m + 1) * m, 0.0), cs(m), sn(m), g(m + 1, 0.0);
      fresh(vs[0]);
      axpby(1.0 / beta, r, 0.0, vs[0]);
      g[0] = beta;
      unsigned j = 0;
      for (; j < m; ++j) {
        fresh(zs[j]);
        fresh(vs[j + 1]);
        P.vmult(zs[j], vs[j]);
        A.vmult(vs[j + 1], zs[j]);
        hcol.resize(j + 1);
        const double hn = orthogonalize(vs, j + 1, vs[j + 1], hcol.data());
        for (unsigned i = 0; i <= j; ++i) H[i * m + j] = hcol[i];
        H[(j + 1) * m + j] = hn;
        if (hn > 0) axpby(1.0 / hn, vs[j + 1], 0.0, vs[j + 1]);
        for (unsigned i = 0; i < j; ++i) {
          const double t = cs[i] * H[i * m + j] + sn[i] * H[(i + 1) * m + j];
          H[(i + 1) * m + j] = -sn[i] * H[i * m + j] + cs[i] * H[(i + 1) * m + j];
          H[i * m + j] = t;
        }
        const double d = std::hypot(H[j * m + j], H[(j + 1) * m + j]);
        cs[j] = H[j * m + j] / d;
        sn[j] = H[(j + 1) * m + j] / d;
        H[j * m + j] = d;
        H[(j + 1) * m + j] = 0.0;
        g[j + 1] = -sn[j] * g[j];
        g[j] = cs[j] * g[j];
        ++steps;
        value = std::abs(g[j + 1]);
        if (verbose && steps % verbose == 0) std::fprintf(stderr, "  FGMRES step %u residual %.3e (start %.3e)\n", steps, value, beta);
        if (value <= tol || steps >= max_steps) {
          ++j;
          break;
        }
      }
      // x += Z y,  H y = g
      std::vector<double> y(j);
      for (int i = int(j) - 1; i >= 0; --i) {
        double s = g[i];
        for (unsigned k = i + 1; k < j; ++k) s -= H[i * m + k] * y[k];
        y[i] = s / H[i * m + i];
      }
      for (unsigned i = 0; i < j; ++i) axpby(y[i], zs[i], 1.0, x);
      if (value <= tol) return;
    }
  }

private:
  unsigned max_steps;
  double abs_tol, reduce;
  unsigned restart, steps = 0;
  double value = 0.0;
  std::vector<V> vs, zs;
  std::vector<double> hcol;
};

// The temporal basis: Lagrange polynomials through get_time_quad's points (fe_time.cc:152-169), evaluated at x
inline std::vector<double> lagrange_values(const std::vector<double> &nodes, double x)
{
  std::vector<double> L(nodes.size(), 1.0);
  for (size_t a = 0; a < nodes.size(); ++a)
    for (size_t m = 0; m < nodes.size(); ++m)
      if (m != a) L[a] *= (x - nodes[m]) / (nodes[a] - nodes[m]);
  return L;
}
inline std::vector<double> time_points(TimeStepType type, unsigned r)
{
  std::vector<double> p(r + 1);
  check(stfem_fe_time_points(type == TimeStepType::CGP ? 0 : 1, int(r), p.data()), "stfem_fe_time_points");
  return p;
}

// A separable function amplitude(t) * prod_d sin(2 pi frequency x_d) - the exact solutions and right-hand sides of the reference's
// convergence tests (include/exact_solution.h:27-81, 147-197) - is evaluated on the device (stfem_integrate_rhs_product,
// stfem_integrate_difference_product): no point list, no host evaluation, no upload
struct ProductFunction {
  double frequency = 1.0;
  std::function<double(double)> amplitude; // of t
  explicit operator bool() const { return bool(amplitude); }
};

// A scalar function of (x, t) evaluated at a list of points: out[i] = f(points[3 i .. 3 i + 2], t)
using PointFunction = std::function<void(double time, const std::vector<double> &points, std::vector<double> &out)>;

// include/time_integrators.h:30-336 for one variable: rhs = rhs_matrix prev_x + time quadrature of the source,
// FGMRES on the slab system.  Alpha / Gamma are the ONE-step temporal matrices (tests/tp_01.cc:123, 525-526).
template <typename Number, typename System, typename RHSSystem, typename Preconditioner> class TimeIntegratorFO {
public:
  using V = BlockVectorT<Number>;
  TimeIntegratorFO(TimeStepType type, unsigned time_degree, const FullMatrix<Number> &Alpha, const FullMatrix<Number> &Gamma,
                   double gmres_tolerance, const System &matrix, const Preconditioner &preconditioner, const RHSSystem &rhs_matrix,
                   const PointFunction &source, unsigned n_timesteps_at_once, bool extrapolate = true, double abstol = 1e-12,
                   unsigned max_steps = 200)
    : type(type), time_degree(time_degree), quad_time(time_points(type, time_degree)), Alpha(Alpha), Gamma(Gamma),
      solver(max_steps, abstol, gmres_tolerance, 100), preconditioner(preconditioner), matrix(matrix), rhs_matrix(rhs_matrix), source(source),
      n_timesteps_at_once(n_timesteps_at_once), nt_dofs(type == TimeStepType::DG ? time_degree + 1 : time_degree), do_extrapolate(extrapolate)
  {
    if (const char *e = std::getenv("STFEM_FGMRES_VERBOSE")) solver.verbose = unsigned(std::atoi(e));
    const Context &c = *matrix_context();
    nq = int(c.degree) + 1; // QGauss(fe degree + 1): the operator's rule (tests/tp_01.cc:95)
  }
  ProductFunction source_product; // if set: the source as a separable function on the device instead of `source` at the points

  // assemble_force (time_integrators.h:73-111): Alpha is diagonal (time quadrature = support points)
  void assemble_force(V &rhs, double time, double time_step) const
  {
    V tmp;
    tmp.reinit(rhs.context(), 1);
    std::vector<double> fq;
    for (unsigned it = 0; it < n_timesteps_at_once; ++it)
      for (unsigned j = 0; j < quad_time.size(); ++j) {
        const double t = time + time_step * it + time_step * quad_time[j];
        if (source_product) {
          check(stfem_integrate_rhs_product(rhs.context()->h, nq, source_product.amplitude(t), source_product.frequency, tmp.handle(), 0, nullptr),
                "stfem_integrate_rhs_product");
        } else {
          if (qpoints.empty()) {
            qpoints.resize(size_t(stfem_n_cells(rhs.context()->h)) * nq * nq * nq * 3);
            check(stfem_quadrature_points(rhs.context()->h, nq, qpoints.data()), "stfem_quadrature_points");
          }
          source(t, qpoints, fq);
          check(stfem_integrate_rhs(rhs.context()->h, nq, fq.data(), tmp.handle(), 0, nullptr), "stfem_integrate_rhs");
        }
        auto add = [&](unsigned block, double w) {
          V view = block_view(rhs, block);
          axpby(w, tmp, 1.0, view);
        };
        if (type == TimeStepType::DG) add(it * nt_dofs + j, Alpha(j, j));
        else if (j == 0)
          for (unsigned i = 0; i < nt_dofs; ++i) add(it * nt_dofs + i, -Gamma(i, 0));
        else add(it * nt_dofs + j - 1, Alpha(j - 1, j - 1));
      }
  }

  // solve (time_integrators.h:300-321); prev_x: one block
  void solve(V &x, const V &prev_x, V &rhs, double time, double time_step)
  {
    TraceRange scope("step");
    const auto t0 = std::chrono::steady_clock::now();
    rhs_matrix.vmult_slice(rhs, prev_x);
    assemble_force(rhs, time, time_step);
    (void)dot(rhs, rhs); // synchronises
    const auto t1 = std::chrono::steady_clock::now();
    for (unsigned b = 0; b < x.n_blocks(); ++b) { // extrapolate (time_integrators.h:184-194)
      V view = block_view(x, b);
      axpby(do_extrapolate ? 1.0 : 0.0, prev_x, 0.0, view);
    }
    solver.solve(matrix, x, rhs, preconditioner);
    (void)dot(x, x);
    assemble_seconds += std::chrono::duration<double>(t1 - t0).count();
    solver_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
  }
  unsigned last_step() const { return solver.last_step(); }
  double assemble_seconds = 0.0, solver_seconds = 0.0; // right-hand side (host evaluation of the source included) / FGMRES

protected:
  std::shared_ptr<Context> matrix_context() const
  {
    V probe;
    matrix.initialize_dof_vector(probe);
    return probe.context();
  }
  TimeStepType type;
  unsigned time_degree;
  std::vector<double> quad_time;
  const FullMatrix<Number> &Alpha, &Gamma;
  SolverFGMRES<Number> solver;
  const Preconditioner &preconditioner;
  const System &matrix;
  const RHSSystem &rhs_matrix;
  PointFunction source;
  unsigned n_timesteps_at_once, nt_dofs;
  bool do_extrapolate;
  int nq = 0;
  mutable std::vector<double> qpoints;
};

// include/time_integrators.h:343-459: the wave equation as a first-order system with the velocity eliminated from the
// slab system (fe_time.h:157-305 builds its temporal matrices); after the solve for u the velocity is recovered block
// by block: v = A^-1 B u + A^-1 Gamma prev (dG: - A^-1 Gamma prev_u; cG: A^-1 Gamma prev_v - A^-1 Zeta prev_u).
// Alpha .. Zeta are the ONE-step heat-type matrices (tests/tp_01.cc:123, 535-545).
template <typename Number, typename System, typename RHSSystem, typename Preconditioner>
class TimeIntegratorWave : public TimeIntegratorFO<Number, System, RHSSystem, Preconditioner> {
  using Base = TimeIntegratorFO<Number, System, RHSSystem, Preconditioner>;

public:
  using V = BlockVectorT<Number>;
  TimeIntegratorWave(TimeStepType type, unsigned time_degree, const FullMatrix<Number> &Alpha, const FullMatrix<Number> &Beta,
                     const FullMatrix<Number> &Gamma, const FullMatrix<Number> &Zeta, double gmres_tolerance, const System &matrix,
                     const Preconditioner &preconditioner, const RHSSystem &rhs_matrix, const RHSSystem &rhs_matrix_v,
                     const PointFunction &source, unsigned n_timesteps_at_once, bool extrapolate = true, unsigned max_steps = 200)
    : Base(type, time_degree, Alpha, Gamma, gmres_tolerance, matrix, preconditioner, rhs_matrix, source, n_timesteps_at_once, extrapolate, 1e-12,
           max_steps),
      rhs_matrix_v(rhs_matrix_v), Alpha_inv(Alpha)
  {
    Alpha_inv.gauss_jordan();
    Alpha_inv.mmult(AixB, Beta);
    Alpha_inv.mmult(AixG, Gamma);
    Alpha_inv.mmult(AixZ, Zeta);
    if (type == TimeStepType::DG) AixG *= Number(-1);
    else AixZ *= Number(-1);
  }

  // prev_u, prev_v: one block each
  void solve(V &u, V &v, V &rhs, const V &prev_u, const V &prev_v, double time, double time_step)
  {
    TraceRange scope("step");
    const auto t0 = std::chrono::steady_clock::now();
    this->rhs_matrix.vmult_slice(rhs, prev_u);
    for (unsigned b = 0; b < u.n_blocks(); ++b) {
      V view = block_view(u, b);
      axpby(this->do_extrapolate ? 1.0 : 0.0, prev_u, 0.0, view);
    }
    rhs_matrix_v.vmult_slice_add(rhs, prev_v);
    this->assemble_force(rhs, time, time_step);
    (void)dot(rhs, rhs);
    const auto t1 = std::chrono::steady_clock::now();
    this->solver.solve(this->matrix, u, rhs, this->preconditioner);
    const unsigned nt_dofs = AixB.m();
    set_zero(v);
    for (unsigned it = 0; it < this->n_timesteps_at_once; ++it) {
      const V pu = it == 0 ? block_view(prev_u, 0) : block_view(u, it * nt_dofs - 1);
      tensorproduct_add(v, AixB, u, it * nt_dofs);
      if (this->type == TimeStepType::DG) tensorproduct_add(v, AixG, pu, it * nt_dofs);
      else {
        const V pv = it == 0 ? block_view(prev_v, 0) : block_view(v, it * nt_dofs - 1);
        tensorproduct_add(v, AixG, pv, it * nt_dofs);
        tensorproduct_add(v, AixZ, pu, it * nt_dofs);
      }
    }
    (void)dot(v, v);
    this->assemble_seconds += std::chrono::duration<double>(t1 - t0).count();
    this->solver_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
  }

private:
  const RHSSystem &rhs_matrix_v;
  FullMatrix<Number> Alpha_inv, AixB, AixG, AixZ;
};

// include/exact_solution.h:503-649: errors of the space-time solution on one slab, QGauss(time degree + 1) in time
// and QGauss(nq_space) per direction in space; u_h(t) from the temporal Lagrange basis (tests/tp_01.cc:404-427)
template <typename Number> class ErrorCalculator {
public:
  using V = BlockVectorT<Number>;
  ErrorCalculator(TimeStepType type, unsigned time_degree, int nq_space, const std::shared_ptr<Context> &ctx, const PointFunction &exact,
                  const PointFunction &exact_gradient)
    : type(type), time_degree(time_degree), nq(nq_space), ctx(ctx), exact(exact), exact_gradient(exact_gradient),
      nodes(time_points(type, time_degree)), tq(time_degree + 1), tw(time_degree + 1)
  {
    check(stfem_gauss_rule(int(time_degree + 1), tq.data(), tw.data()), "stfem_gauss_rule");
    numeric.reinit(ctx, 1);
  }
  ProductFunction exact_product; // if set: the exact solution (and its gradient) as a separable function on the device
  // returns {L2^2 contribution, Linfty, H1-semi^2 contribution} of the slab [time, time + n_steps * time_step]
  std::array<double, 3> evaluate_error(double time, double time_step, const V &x, const V &prev_x, unsigned n_time_steps_at_once)
  {
    std::array<double, 3> err{0.0, -1.0, 0.0};
    const unsigned nt_dofs = type == TimeStepType::DG ? time_degree + 1 : time_degree;
    std::vector<double> ue, ge;
    for (unsigned it = 0; it < n_time_steps_at_once; ++it)
      for (unsigned q = 0; q < tq.size(); ++q) {
        const double t = time + time_step * it + tq[q] * time_step;
        const std::vector<double> L = lagrange_values(nodes, tq[q]);
        // evaluate_numerical_solution: DG: sum_i L_i x_i; CGP: L_0 prev + sum_{i>=1} L_i x_{i-1}
        set_zero(numeric);
        if (type == TimeStepType::DG) {
          for (unsigned i = 0; i < nt_dofs; ++i) axpby(L[i], block_view(x, it * nt_dofs + i), 1.0, numeric);
        } else {
          if (it == 0) axpby(L[0], prev_x, 1.0, numeric);
          else axpby(L[0], block_view(x, nt_dofs * it - 1), 1.0, numeric);
          for (unsigned i = 1; i <= time_degree; ++i) axpby(L[i], block_view(x, it * nt_dofs + i - 1), 1.0, numeric);
        }
        double out[3];
        if (exact_product) {
          check(stfem_integrate_difference_product(ctx->h, nq, numeric.handle(), 0, exact_product.amplitude(t), exact_product.frequency, out, nullptr),
                "stfem_integrate_difference_product");
        } else {
          if (qpoints.empty()) {
            qpoints.resize(size_t(stfem_n_cells(ctx->h)) * nq * nq * nq * 3);
            check(stfem_quadrature_points(ctx->h, nq, qpoints.data()), "stfem_quadrature_points");
          }
          exact(t, qpoints, ue);
          exact_gradient(t, qpoints, ge);
          check(stfem_integrate_difference(ctx->h, nq, numeric.handle(), 0, ue.data(), ge.data(), out, nullptr), "stfem_integrate_difference");
        }
        err[0] += time_step * tw[q] * out[0];
        err[1] = std::max(err[1], out[1]);
        err[2] += time_step * tw[q] * out[2];
      }
    return err;
  }

private:
  TimeStepType type;
  unsigned time_degree;
  int nq;
  std::shared_ptr<Context> ctx;
  PointFunction exact, exact_gradient;
  std::vector<double> nodes, tq, tw, qpoints;
  V numeric;
};

} // namespace stfem
