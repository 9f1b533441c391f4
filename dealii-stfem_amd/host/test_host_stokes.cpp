// C++ caller of the Stokes mirror (in the style of tests/tp_03stokes.cc): builds the operator and the
// space-time system, applies vmult to seeded vectors, writes inputs and results for the Python test.
//   test_host_stokes ncx ncy ncz type r nsteps viscosity out.bin [weak_mask]
// weak_mask: boundary ids with Nitsche conditions (the other faces keep their strong constraints); the file then also
// holds StokesNitscheMatrixFreeOperator::vmult for the Dirichlet function g below.
#include "stfem/stokes.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

using namespace stfem;

int main(int argc, char **argv)
{
  if (argc != 9 && argc != 10) return 2;
  try {
    Mesh mesh;
    for (int d = 0; d < 3; ++d) mesh.ncell[d] = std::atoi(argv[1 + d]);
    mesh.distort_random(0.1, 99);
    const TimeStepType type = std::atoi(argv[4]) == 0 ? TimeStepType::CGP : TimeStepType::DG;
    const unsigned r = unsigned(std::atoi(argv[5])), ns = unsigned(std::atoi(argv[6]));
    const double nu = std::atof(argv[7]);
    const int weak = argc == 10 ? std::atoi(argv[9]) : 0;
    std::set<boundary_id> weak_ids;
    for (unsigned f = 0; f < 6; ++f)
      if (weak >> f & 1) weak_ids.insert(f);
    mesh.dirichlet_mask = 63 & ~weak;
    StokesMatrixFreeOperator<3, double> K(mesh, 2, nu, weak_ids);
    const auto w = get_fe_time_weights_stokes<double>(type, r, 1.0 / 32, ns);
    const unsigned nt = type == TimeStepType::CGP ? r : r + 1;
    BlockSlice slice(ns, 2, nt);
    SystemMatrixStokes<3, double> A(K, w[0], w[1], slice);
    std::vector<StokesVector> x, y;
    A.initialize_dof_vector(x);
    A.initialize_dof_vector(y);
    FILE *f = std::fopen(argv[8], "wb");
    if (!f) return 3;
    const unsigned long long nb = x.size();
    std::fwrite(&nb, sizeof nb, 1, f);
    for (unsigned b = 0; b < nb; ++b) {
      std::vector<double> h(x[b].size());
      std::mt19937_64 rng(4321 + b);
      for (double &v : h) v = double(rng() >> 11) * (2.0 / 9007199254740992.0) - 1.0;
      x[b].copy_from_host(h);
      const unsigned long long n = h.size();
      std::fwrite(&n, sizeof n, 1, f);
      std::fwrite(h.data(), sizeof(double), n, f);
    }
    A.vmult(y, x);
    for (unsigned b = 0; b < nb; ++b) {
      const auto h = y[b].copy_to_host();
      std::fwrite(h.data(), sizeof(double), h.size(), f);
    }
    if (weak) { // tests/tp_03stokes.cc:189-207, 876-878: the right-hand side of the weakly imposed Dirichlet data
      StokesNitscheMatrixFreeOperator<3, double> N(K);
      N.set_dirichlet_functions([](const std::array<double, 3> &x) {
        return std::array<double, 3>{{std::sin(x[0] + 2 * x[1]), x[2] * x[2] - x[0], std::cos(x[1] * x[2])}};
      });
      std::vector<StokesVector> rhs;
      N.initialize_dof_vector(rhs);
      N.vmult(rhs);
      for (const auto &v : rhs) {
        const auto h = v.copy_to_host();
        std::fwrite(h.data(), sizeof(double), h.size(), f);
      }
    }
    std::fclose(f);
    int thrown = 0;
    try { A.vmult(x, x); } catch (const Error &e) { thrown += e.status == STFEM_ERR_ALIAS; }
    std::printf("m=%llu blocks=%llu exceptions=%d\n", A.m(), nb, thrown);
    return thrown == 1 ? 0 : 4;
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
}
