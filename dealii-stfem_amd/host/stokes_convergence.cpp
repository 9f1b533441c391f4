// The reference's space-time convergence test of the instationary Stokes problem (tests/tp_03stokes.cc) in 3D on the device:
// FE_Q(2)^3 x FE_Q(1) in space (BASELINE configs[4]), dG(k) / cG(k) in time, tau = 2^-(refinement + 1), one time step per solve,
// homogeneous Dirichlet velocity on the whole boundary, pressure with zero mean, FGMRES (1e-12) preconditioned by relaxation sweeps
// of the two-variable cell-patch Vanka smoother (the smoother of the reference's Stokes multigrid levels, tests/tp_03stokes.cc:714-726).
// The reference's exact solution (include/exact_solution.h:199-325) is two-dimensional; this driver uses its 3D analogue: the
// velocity is the curl of psi e_z, psi = sin t (sin pi x sin pi y sin pi z)^2, the pressure sin t cos pi x cos pi y cos pi z.
// With mg=<levels> the preconditioner is one V-cycle of the geometric multigrid of the reference's Stokes runs (GMGStokes in
// host/stfem/stokes_solver.h: <levels> space levels, relaxation sweeps of the Vanka smoother on every level, 2^(levels - 1 - l) smoothing
// steps on level l); with stmg=1 in addition the levels in time of the reference's sequence (get_mg_sequence as tests/tp_03stokes.cc:294-312
// calls it: the temporal degree goes down to 1 (cG) / 0 (dG) by bisection, space_or_time unless coarsening=space_and_time).  The errors do
// not depend on the preconditioner, the iteration counts do.
// Usage: stokes_convergence <type 0 = cG | 1 = dG> <k> <refinement> [vanka sweeps = 3] [omega = 0: estimated] [viscosity = 1] [cells per direction]
//                           [end_time = 1] [mg=<levels>] [stmg=1] [coarsening=space_and_time] [dg=1]
// Prints: cells u-dofs p-dofs t-dofs  u:Linf-Linf  u:L2-L2  u:L2-H1semi  p:L2-L2  gmres-iterations-per-solve
#include "stfem/stokes_solver.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

using namespace stfem;

namespace {
const double PI = 3.14159265358979323846;
inline double A(double s) { const double q = std::sin(PI * s); return q * q; }
inline double dA(double s) { return PI * std::sin(2 * PI * s); }
inline double d2A(double s) { return 2 * PI * PI * std::cos(2 * PI * s); }
inline double B(double s) { return 0.5 * std::sin(2 * PI * s); }
inline double dB(double s) { return PI * std::cos(2 * PI * s); }
inline double d2B(double s) { return -4 * PI * PI * B(s); }
// the analytic functions are evaluated at up to 10^7 points per call (27 quadrature points per cell): the point loop in slices on
// the host's cores (the reference evaluates its Functions inside the threaded cell loops of deal.II)
template <typename Body> void for_points(size_t n, Body &&body)
{
  const unsigned nthreads = n < 65536 ? 1u : std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  if (nthreads == 1) {
    body(size_t(0), n);
    return;
  }
  std::vector<std::thread> pool;
  const size_t chunk = (n + nthreads - 1) / nthreads;
  for (unsigned t = 0; t < nthreads; ++t) {
    const size_t lo = std::min(n, t * chunk), hi = std::min(n, lo + chunk);
    if (lo < hi) pool.emplace_back([&body, lo, hi] { body(lo, hi); });
  }
  for (auto &th : pool) th.join();
}
} // namespace

int main(int argc_all, char **argv_all)
{
  unsigned mg_levels = 0;
  bool stmg = false, space_and_time = false;
  bool dg_pressure = false; // dg=1: FE_DGP(1) pressure, the reference's default (tests/json/stokes.json: dGPressure = true)
  std::vector<char *> pos;
  for (int i = 0; i < argc_all; ++i) {
    if (i > 0 && std::strncmp(argv_all[i], "mg=", 3) == 0) mg_levels = unsigned(std::atoi(argv_all[i] + 3));
    else if (i > 0 && std::strncmp(argv_all[i], "dg=", 3) == 0) dg_pressure = std::atoi(argv_all[i] + 3) != 0;
    else if (i > 0 && std::strncmp(argv_all[i], "stmg=", 5) == 0) stmg = std::atoi(argv_all[i] + 5) != 0;
    else if (i > 0 && std::strcmp(argv_all[i], "coarsening=space_and_time") == 0) space_and_time = true;
    else pos.push_back(argv_all[i]);
  }
  const int argc = int(pos.size());
  char **argv = pos.data();
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s type k refinement [sweeps] [omega] [viscosity] [cells] [end_time]\n", argv[0]);
    return 2;
  }
  const auto type = std::atoi(argv[1]) == 0 ? TimeStepType::CGP : TimeStepType::DG;
  const unsigned k = std::atoi(argv[2]), refinement = std::atoi(argv[3]);
  const unsigned sweeps = argc > 4 ? std::atoi(argv[4]) : 3;
  const double omega_arg = argc > 5 ? std::atof(argv[5]) : 0.0; // 0: estimated (deal.II's PreconditionRelaxation with relaxation = 0)
  const double nu = argc > 6 ? std::atof(argv[6]) : 1.0;
  const int n = argc > 7 ? std::atoi(argv[7]) : 1 << refinement;
  const double tau = std::ldexp(1.0, -int(refinement + 1)), end_time = argc > 8 ? std::atof(argv[8]) : 1.0;
  try {
    Mesh mesh;
    mesh.ncell[0] = mesh.ncell[1] = mesh.ncell[2] = n;
    StokesMatrixFreeOperator<3, double> K(mesh, 2, nu, std::set<boundary_id>(), std::set<boundary_id>(), 20.0, 10.0, 0.0, 0.0, 0.0, dg_pressure);
    auto spaces = std::make_shared<StokesSpaces>(mesh, K.handle());
    const unsigned nt = type == TimeStepType::CGP ? k : k + 1;
    const BlockSlice slice(1, 2, nt), slice1(1, 2, 1);
    const auto w = get_fe_time_weights_stokes<double>(type, k, tau, 1); // Alpha, Beta, Gamma, Zeta (fe_time.h:1242-1285)
    auto [Alpha_1, Beta_1, Gamma_1, Zeta_1] = get_fe_time_weights<double>(type, k, tau, 1);
    (void)Beta_1; (void)Zeta_1;
    SystemMatrixStokes<3, double> matrix(K, w[0], w[1], slice);
    // right-hand-side matrices (tests/tp_03stokes.cc:243-244): cG: Gamma on K_S, Zeta on M; dG: Gamma on M
    FullMatrix<double> zero(w[2].m(), w[2].n());
    const bool cgp = type == TimeStepType::CGP;
    SystemMatrixStokes<3, double> rhs_matrix(K, cgp ? w[2] : zero, cgp ? w[3] : w[2], slice);
    StokesSystem<3, double> system(matrix, spaces, K.handle(), slice);
    PreconditionVankaStokes<double> vanka(K, w[0], w[1], slice);
    const double omega = omega_arg != 0.0 ? omega_arg : estimate_relaxation_stokes(system, vanka);
    PreconditionRelaxationStokes<StokesSystem<3, double>> preconditioner(system, vanka, omega, sweeps);

    const VectorPointFunction force = [&](double t, const std::vector<double> &p, std::array<std::vector<double>, 3> &out) {
      const size_t np = p.size() / 3;
      const double st = std::sin(t), ct = std::cos(t);
      for (auto &o : out) o.resize(np);
      for_points(np, [&](size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; ++i) {
        const double x = p[3 * i], y = p[3 * i + 1], z = p[3 * i + 2];
        const double lap1 = d2A(x) * B(y) * A(z) + A(x) * d2B(y) * A(z) + A(x) * B(y) * d2A(z);
        const double lap2 = d2B(x) * A(y) * A(z) + B(x) * d2A(y) * A(z) + B(x) * A(y) * d2A(z);
        const double sx = std::sin(PI * x), sy = std::sin(PI * y), sz = std::sin(PI * z), cx = std::cos(PI * x), cy = std::cos(PI * y), cz = std::cos(PI * z);
        out[0][i] = 2 * PI * (ct * A(x) * B(y) * A(z) - nu * st * lap1) - PI * st * sx * cy * cz;
        out[1][i] = -2 * PI * (ct * B(x) * A(y) * A(z) - nu * st * lap2) - PI * st * cx * sy * cz;
        out[2][i] = -PI * st * cx * cy * sz;
      }
      });
    };
    auto exact_u = [&](int c) {
      return PointFunction([c](double t, const std::vector<double> &p, std::vector<double> &out) {
        out.resize(p.size() / 3);
        const double a = 2 * PI * std::sin(t);
        for_points(out.size(), [&](size_t lo, size_t hi) {
          for (size_t i = lo; i < hi; ++i) {
            const double x = p[3 * i], y = p[3 * i + 1], z = p[3 * i + 2];
            out[i] = c == 0 ? a * A(x) * B(y) * A(z) : (c == 1 ? -a * B(x) * A(y) * A(z) : 0.0);
          }
        });
      });
    };
    auto exact_grad_u = [&](int c) {
      return PointFunction([c](double t, const std::vector<double> &p, std::vector<double> &out) {
        out.assign(p.size(), 0.0);
        const double a = 2 * PI * std::sin(t);
        for_points(p.size() / 3, [&](size_t lo, size_t hi) {
          for (size_t i = lo; i < hi; ++i) {
            const double x = p[3 * i], y = p[3 * i + 1], z = p[3 * i + 2];
            if (c == 0) { out[3 * i] = a * dA(x) * B(y) * A(z); out[3 * i + 1] = a * A(x) * dB(y) * A(z); out[3 * i + 2] = a * A(x) * B(y) * dA(z); }
            if (c == 1) { out[3 * i] = -a * dB(x) * A(y) * A(z); out[3 * i + 1] = -a * B(x) * dA(y) * A(z); out[3 * i + 2] = -a * B(x) * A(y) * dA(z); }
          }
        });
      });
    };
    const PointFunction exact_p = [](double t, const std::vector<double> &p, std::vector<double> &out) {
      out.resize(p.size() / 3);
      for_points(out.size(), [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) out[i] = std::sin(t) * std::cos(PI * p[3 * i]) * std::cos(PI * p[3 * i + 1]) * std::cos(PI * p[3 * i + 2]);
      });
    };

    // the preconditioner behind one interface: relaxation sweeps on the finest level, or one V-cycle
    std::unique_ptr<GMGStokes<3>> gmg;
    if (mg_levels > 0) {
      GMGStokes<3>::AdditionalData ad;
      ad.smoothing_degree = sweeps;
      ad.relaxation = omega_arg;
      if (stmg) {
        const auto poly_time = get_poly_mg_sequence(k, type == TimeStepType::CGP ? 1u : 0u, PolynomialCoarseningSequenceType::bisect);
        const auto seq = get_mg_sequence(mg_levels, poly_time, std::vector<unsigned>{2}, 1, 1, MGType::tau,
                                         space_and_time ? CoarseningType::space_and_time : CoarseningType::space_or_time, false, false, true);
        std::fprintf(stderr, "levels:");
        for (MGType t : seq) std::fprintf(stderr, " %c", char(t));
        std::fprintf(stderr, "\n");
        gmg = std::make_unique<GMGStokes<3>>(mesh, seq, poly_time, type, tau, 1, nu, ad, std::set<boundary_id>(), dg_pressure);
      } else gmg = std::make_unique<GMGStokes<3>>(mesh, mg_levels, nu, w[0], w[1], slice, ad, std::set<boundary_id>(), dg_pressure);
      for (unsigned l = 0; l < gmg->n_levels(); ++l) std::fprintf(stderr, "level %u: relaxation %.4f\n", l, gmg->relaxation(l));
    } else
      std::fprintf(stderr, "relaxation %.4f\n", omega);
    struct Prec {
      const PreconditionRelaxationStokes<StokesSystem<3, double>> *relax;
      const GMGStokes<3> *gmg;
      void vmult(StokesBlockVector &dst, const StokesBlockVector &src) const
      {
        if (gmg) gmg->vmult(dst, src);
        else relax->vmult(dst, src);
      }
    } prec{&preconditioner, gmg.get()};
    TimeIntegratorStokes<3, StokesSystem<3, double>, Prec> step(type, k, Alpha_1, Gamma_1, 1e-12, system, prec, rhs_matrix, force, true);
    // ErrorCalculator (exact_solution.h:503-649): QGauss(k + 1) in time; QGauss(3) per direction for the velocity components, QGauss(2) for the pressure
    std::vector<ErrorCalculator<double>> err_u;
    for (int c = 0; c < 3; ++c) err_u.emplace_back(type, k, 3, spaces->q2, exact_u(c), exact_grad_u(c));
    PressureErrorCalculator err_p(type, k, 2, spaces, exact_p);

    StokesBlockVector x, rhs, prev;
    x.reinit(spaces, K.handle(), slice);
    rhs.reinit(spaces, K.handle(), slice);
    prev.reinit(spaces, K.handle(), slice1); // u(0) = 0, p(0) = 0
    const size_t nu_dofs = size_t(stfem_stokes_n_velocity_dofs(K.handle()));
    double time = 0.0, l2 = 0.0, l8 = -1.0, h1 = 0.0, l2p = 0.0;
    unsigned solves = 0, iterations = 0;
    double solve_seconds = 0.0;
    while (time < end_time - 1e-12) {
      const auto t0 = std::chrono::steady_clock::now();
      step.solve(x, prev, rhs, time, tau);
      (void)dot(x, x); // synchronises
      solve_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      iterations += step.last_step();
      ++solves;
      // component c of the time dofs as a block vector of the scalar velocity space; the pressure blocks likewise
      for (int c = 0; c < 3; ++c) {
        std::vector<void *> ptrs(nt);
        for (unsigned a = 0; a < nt; ++a) ptrs[a] = x.blocks()[slice.index(0, 0, a)].data() + c * nu_dofs;
        BlockVectorT<double> xc, pc;
        xc.wrap(spaces->q2, ptrs.data(), nt);
        void *pp[1] = {prev.blocks()[0].data() + c * nu_dofs};
        pc.wrap(spaces->q2, pp, 1);
        const auto e = err_u[c].evaluate_error(time, tau, xc, pc, 1);
        l2 += e[0];
        l8 = std::max(l8, e[1]);
        h1 += e[2];
      }
      {
        std::vector<void *> ptrs(nt);
        for (unsigned a = 0; a < nt; ++a) ptrs[a] = x.blocks()[slice.index(0, 1, a)].data();
        BlockVectorT<double> xp, pp;
        xp.wrap(spaces->q1, ptrs.data(), nt);
        void *q[1] = {prev.blocks()[1].data()};
        pp.wrap(spaces->q1, q, 1);
        l2p += err_p.evaluate_error(time, tau, xp, pp)[0];
      }
      axpby(1.0, x.view(slice.index(0, 0, nt - 1)), 0.0, prev.view(0));
      axpby(1.0, x.view(slice.index(0, 1, nt - 1)), 0.0, prev.view(1));
      time += tau;
    }
    std::fprintf(stderr, "%u slab solves: %.3f s (right-hand side on the host + FGMRES), FGMRES alone %.3f s for %u iterations = %.2f ms per iteration\n", solves,
                 solve_seconds, step.solver_seconds(), iterations, 1e3 * step.solver_seconds() / std::max(1u, iterations));
    if (gmg) gmg->print_timing(stderr);
    std::printf("%d %lld %lld %u %.12e %.12e %.12e %.12e %.2f\n", n * n * n, 3ll * (long long)nu_dofs, (long long)stfem_stokes_n_pressure_dofs(K.handle()), nt, l8,
                std::sqrt(l2), std::sqrt(h1), std::sqrt(l2p), double(iterations) / solves);
    return 0;
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
}
