// The reference's space-time convergence test of the heat equation (tests/tp_01.cc with space_time_conv_test,
// ProblemType::heat) in 3D on the device: u = sin(2 pi f t) prod sin(2 pi f x_d) on the unit cube, FE_Q(k + 1) in
// space, dG(k) / cG(k) in time, tau = 2^-(refinement + 1), n_timesteps_at_once steps per solve, FGMRES
// (200 steps, restart 100, 1e-12) preconditioned by relaxation sweeps of the cell-patch Vanka smoother or, with mg=1, by the
// reference's own preconditioner, one V-cycle of the space-time multigrid (the errors do not depend on the preconditioner,
// the iteration counts do).
// With mg=1 the preconditioner is the reference's own: one V-cycle of the space-time multigrid (host/stfem/stmg.h), levels as
// tests/tp_01.cc:170-200 derives them.  Options (key=value, anywhere): mg=0|1, mg_float=0|1 (multigrid in fp32, stmg.h:1330-1343),
// distort=<vertex jitter in h>, coarsening=space_or_time|space_and_time, pmg=0|1, kmin=<lowest temporal degree>, relaxation=<omega, 0 = estimated>, variable=0|1, steps=<n>
// Usage: heat_convergence <type 0 = cG | 1 = dG> <k> <refinement> <n_timesteps_at_once> [vanka sweeps = 2, 0 = none] [omega = 0.5]
//                         [fe_degree = k + 1] [cells per direction = 2^refinement] [end_time = 1] [FGMRES steps = 200]
// Prints: cells s-dofs t-dofs Linf-Linf L2-L2 L2-H1semi gmres-iterations-per-solve  (and timings on stderr)
#include "stfem/stmg.h"
#include "stfem/time_integrators.h"

#include <cstring>
#include <map>

#include <cstdio>
#include <cstdlib>

using namespace stfem;
using Number = double;

int main(int argc_all, char **argv_all)
{
  std::map<std::string, std::string> opt;
  std::vector<char *> pos;
  for (int i = 0; i < argc_all; ++i) {
    const char *eq = std::strchr(argv_all[i], '=');
    if (i > 0 && eq) opt[std::string(argv_all[i], size_t(eq - argv_all[i]))] = eq + 1;
    else pos.push_back(argv_all[i]);
  }
  const int argc = int(pos.size());
  char **argv = pos.data();
  auto option = [&](const char *key, const char *dflt) { return opt.count(key) ? opt[key] : std::string(dflt); };
  if (argc < 5) {
    std::fprintf(stderr, "usage: %s type k refinement n_timesteps_at_once [sweeps] [omega]\n", argv[0]);
    return 2;
  }
  const auto type = std::atoi(argv[1]) == 0 ? TimeStepType::CGP : TimeStepType::DG;
  const unsigned k = std::atoi(argv[2]), refinement = std::atoi(argv[3]), nsteps = std::atoi(argv[4]);
  const unsigned sweeps = argc > 5 ? std::atoi(argv[5]) : 2;
  const double omega = argc > 6 ? std::atof(argv[6]) : 0.5;
  const unsigned fe_degree = argc > 7 ? std::atoi(argv[7]) : k + 1; // tests/tp_01.cc:76
  const int n = argc > 8 ? std::atoi(argv[8]) : 1 << refinement;    // subdivided_hyper_rectangle with one subdivision, refined globally
  const double tau = std::ldexp(1.0, -int(refinement + 1)), end_time = argc > 9 ? std::atof(argv[9]) : 1.0, f = 1.0, PI = 3.14159265358979323846;
  const unsigned max_steps = argc > 10 ? std::atoi(argv[10]) : 200;
  try {
    Mesh mesh;
    mesh.ncell[0] = mesh.ncell[1] = mesh.ncell[2] = n;
    const double distort = std::atof(option("distort", "0").c_str()); // GridTools::distort_random (tests/tp_01.cc:89-90): general-geometry path, per-cell Vanka blocks
    if (distort > 0) mesh.distort_random(distort);
    MatrixFreeOperatorScalar<3, Number> K_mf(mesh, fe_degree, 0.0, 1.0), M_mf(K_mf, 1.0, 0.0);
    auto [Alpha, Beta, Gamma, Zeta] = get_fe_time_weights<Number>(type, k, tau, nsteps);
    auto [Alpha_1, Beta_1, Gamma_1, Zeta_1] = get_fe_time_weights<Number>(type, k, tau, 1);
    (void)Beta_1; (void)Zeta_1;
    using SystemN = SystemMatrix<3, Number, MatrixFreeOperatorScalar<3, Number>>;
    SystemN matrix(K_mf, M_mf, Alpha, Beta);
    const bool cgp = type == TimeStepType::CGP;
    FullMatrix<Number> zero(Gamma.m(), 1);
    SystemN rhs_matrix(K_mf, M_mf, cgp ? Gamma : zero, cgp ? Zeta : Gamma); // tests/tp_01.cc:160-166

    auto product = [&](double amp, const std::vector<double> &pts, std::vector<double> &out) {
      out.resize(pts.size() / 3);
      for (size_t i = 0; i < out.size(); ++i)
        out[i] = amp * std::sin(2 * PI * f * pts[3 * i]) * std::sin(2 * PI * f * pts[3 * i + 1]) * std::sin(2 * PI * f * pts[3 * i + 2]);
    };
    // include/exact_solution.h:27-81
    PointFunction exact = [&](double t, const std::vector<double> &pts, std::vector<double> &out) { product(std::sin(2 * PI * f * t), pts, out); };
    PointFunction source = [&](double t, const std::vector<double> &pts, std::vector<double> &out) {
      product(3 * 4 * PI * PI * f * f * std::sin(2 * PI * f * t) + 2 * PI * f * std::cos(2 * PI * f * t), pts, out);
    };
    PointFunction exact_grad = [&](double t, const std::vector<double> &pts, std::vector<double> &out) {
      out.resize(pts.size());
      const double tv = 2 * PI * f * std::sin(2 * PI * f * t);
      for (size_t i = 0; i < pts.size() / 3; ++i)
        for (int d = 0; d < 3; ++d) {
          double g = tv;
          for (int e = 0; e < 3; ++e) g *= d == e ? std::cos(2 * PI * f * pts[3 * i + e]) : std::sin(2 * PI * f * pts[3 * i + e]);
          out[3 * i + d] = g;
        }
    };

    BlockVectorT<Number> x, rhs, prev_x;
    matrix.initialize_dof_vector(x);
    matrix.initialize_dof_vector(rhs);
    prev_x.reinit(K_mf.context(), 1); // u(0) = 0 (VectorTools::interpolate of the exact solution at t = 0)
    {
      std::vector<double> pts(3 * prev_x.block_size()), u0;
      check(stfem_support_points(K_mf.context()->h, pts.data()), "stfem_support_points");
      exact(0.0, pts, u0);
      prev_x.copy_from_host({u0});
    }
    ErrorCalculator<Number> error_calculator(type, k, int(k + 1), K_mf.context(), exact, exact_grad); // exact_solution.h:524-526

    double l2 = 0.0, l8 = -1.0, h1 = 0.0, time = 0.0, rhs_s = 0.0, solve_s = 0.0;
    unsigned total_its = 0, solves = 0;
    // host_functions=1: the source and the exact solution evaluated on the host at the quadrature points (the general interface: any function);
    // default: as separable functions on the device
    const bool on_device = option("host_functions", "0") != "1";
    if (on_device) error_calculator.exact_product = ProductFunction{f, [&](double t) { return std::sin(2 * PI * f * t); }};
    auto run = [&](auto &step) {
      if (on_device) step.source_product = ProductFunction{f, [&](double t) { return 3 * 4 * PI * PI * f * f * std::sin(2 * PI * f * t) + 2 * PI * f * std::cos(2 * PI * f * t); }};
      while (time < end_time - 1e-12) {
        step.solve(x, prev_x, rhs, time, tau);
        rhs_s = step.assemble_seconds;
        solve_s = step.solver_seconds;
        total_its += step.last_step();
        ++solves;
        const auto e = error_calculator.evaluate_error(time, tau, x, prev_x, nsteps);
        l2 += e[0];
        l8 = std::max(l8, e[1]);
        h1 += e[2];
        axpby(1.0, block_view(x, x.n_blocks() - 1), 0.0, prev_x);
        time += nsteps * tau;
      }
    };
    if (option("mg", "0") == "1") {
      // tests/tp_01.cc:170-200: one mesh level per halving down to one cell, temporal degrees by bisection, tau levels down to one step
      unsigned n_sp_lvl = 1;
      for (int c = n; c % 2 == 0; c /= 2) ++n_sp_lvl;
      const unsigned kmin = std::min<unsigned>(k, std::atoi(option("kmin", "1").c_str()));
      const auto poly_time = get_poly_mg_sequence(k, kmin, PolynomialCoarseningSequenceType::bisect);
      std::vector<unsigned> poly_space;
      for (unsigned q : poly_time) poly_space.push_back(q + (fe_degree - k)); // get_fe_pmg_sequence: FE_Q(time degree + 1)
      const bool use_pmg = option("pmg", "0") == "1";
      const auto ctype = option("coarsening", "space_or_time") == "space_and_time" ? CoarseningType::space_and_time : CoarseningType::space_or_time;
      const auto mg_type_level = get_mg_sequence(n_sp_lvl, poly_time, poly_space, nsteps, 1, MGType::tau, ctype, false, use_pmg, true);
      PreconditionerGMGAdditionalData mg_data;
      mg_data.relaxation = std::atof(option("relaxation", "0").c_str());
      mg_data.variable = option("variable", "1") == "1";
      mg_data.smoothing_steps = std::atoi(option("steps", "1").c_str());
      if (option("smoother", "relaxation") == "chebyshev") mg_data.smoother = SupportedSmoothers::Chebyshev; // steps = its degree
      std::fprintf(stderr, "levels:");
      for (auto m : mg_type_level) std::fprintf(stderr, " %c", char(m));
      std::fprintf(stderr, "\n");
      auto with = [&](auto number_tag) {
        using NP = decltype(number_tag);
        STMGHierarchy<3, NP> mg(mesh, fe_degree, poly_space, type, tau, nsteps, mg_type_level, poly_time, mg_data, ctype, false, true);
        for (unsigned l = 0; l < mg.operators.size(); ++l)
          std::fprintf(stderr, "level %u: %llu x %u dofs, smoother %u, relaxation %.4f\n", l, (unsigned long long)mg.K[l]->m(), mg.fetw[l][0].m(),
                       mg.gmg->smoother_types()[l], mg.gmg->relaxation(l));
        using P = GMG<3, NP, typename STMGHierarchy<3, NP>::System>;
        TimeIntegratorFO<Number, SystemN, SystemN, P> step(type, k, Alpha_1, Gamma_1, 1e-12, matrix, *mg.gmg, rhs_matrix, source, nsteps, true, 1e-12, max_steps);
        run(step);
      };
      if (option("mg_float", "0") == "1") with(float());
      else with(double());
    } else if (sweeps > 0) {
      PreconditionVanka<Number> vanka(K_mf, Alpha, Beta);
      PreconditionRelaxation<Number, SystemN> precond(matrix, vanka, omega, sweeps);
      TimeIntegratorFO<Number, SystemN, SystemN, decltype(precond)> step(type, k, Alpha_1, Gamma_1, 1e-12, matrix, precond, rhs_matrix, source, nsteps, true, 1e-12, max_steps);
      run(step);
    } else {
      PreconditionIdentity precond;
      TimeIntegratorFO<Number, SystemN, SystemN, PreconditionIdentity> step(type, k, Alpha_1, Gamma_1, 1e-12, matrix, precond, rhs_matrix, source, nsteps, true, 1e-12, max_steps);
      run(step);
    }
    std::fprintf(stderr, "%u slab solves: right-hand sides %.3f s (source evaluated on the %s), FGMRES %.3f s for %u iterations = %.2f ms per iteration\n",
                 solves, rhs_s, on_device ? "device" : "host", solve_s, total_its, 1e3 * solve_s / std::max(1u, total_its));
    std::printf("%d %llu %u %.12e %.12e %.12e %.2f\n", n * n * n, (unsigned long long)K_mf.m(), x.n_blocks(), l8, std::sqrt(l2), std::sqrt(h1),
                double(total_its) / solves);
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
