// C++ caller of the Stokes multigrid mirror (GMGStokes, host/stfem/stokes_solver.h): one V-cycle applied to seeded vectors, written
// out for the Python test that compares it with the numpy V-cycle of oracle/stmg_oracle.py on dense level matrices.
//   test_host_stokes_mg n levels type r viscosity smoothing_degree omega variable out.bin [dg_pressure [sequence steps]]
// sequence: the level transitions coarse to fine, e.g. "hk" (h, then k: the temporal degree rises by one per k towards the finest
// level, a t doubles the time steps per slab); `levels` is ignored then.  steps: time steps per slab on the finest level.
#include "stfem/stokes_solver.h"

#include <cstdio>
#include <cstdlib>
#include <memory>
#include <random>

using namespace stfem;

int main(int argc, char **argv)
{
  if (argc != 10 && argc != 11 && argc != 13) return 2;
  try {
    Mesh mesh;
    mesh.ncell[0] = mesh.ncell[1] = mesh.ncell[2] = std::atoi(argv[1]);
    const unsigned levels = unsigned(std::atoi(argv[2]));
    const TimeStepType type = std::atoi(argv[3]) == 0 ? TimeStepType::CGP : TimeStepType::DG;
    const unsigned r = unsigned(std::atoi(argv[4]));
    const double nu = std::atof(argv[5]);
    GMGStokes<3>::AdditionalData ad;
    ad.smoothing_degree = unsigned(std::atoi(argv[6]));
    ad.relaxation = std::atof(argv[7]);
    ad.variable = std::atoi(argv[8]) != 0;
    const unsigned nt = type == TimeStepType::CGP ? r : r + 1;
    const unsigned steps = argc == 13 ? unsigned(std::atoi(argv[12])) : 1;
    const BlockSlice slice(steps, 2, nt);
    const auto w = get_fe_time_weights_stokes<double>(type, r, 1.0 / 16, steps);
    const bool dg = argc >= 11 && std::atoi(argv[10]) != 0;
    std::unique_ptr<GMGStokes<3>> gmg_ptr;
    if (argc == 13) {
      std::vector<MGType> seq;
      unsigned n_k = 0;
      for (const char *c = argv[11]; *c; ++c) {
        seq.push_back(MGType(*c));
        n_k += *c == 'k';
      }
      std::vector<unsigned> degrees;
      for (unsigned d = r - n_k; d <= r; ++d) degrees.push_back(d);
      gmg_ptr = std::make_unique<GMGStokes<3>>(mesh, seq, degrees, type, 1.0 / 16, steps, nu, ad, std::set<boundary_id>(), dg);
    } else gmg_ptr = std::make_unique<GMGStokes<3>>(mesh, levels, nu, w[0], w[1], slice, ad, std::set<boundary_id>(), dg);
    GMGStokes<3> &gmg = *gmg_ptr;
    StokesBlockVector x, y;
    gmg.finest_system().initialize_dof_vector(x);
    gmg.finest_system().initialize_dof_vector(y);
    FILE *f = std::fopen(argv[9], "wb");
    if (!f) return 3;
    const unsigned long long nb = x.n_blocks();
    std::fwrite(&nb, sizeof nb, 1, f);
    for (unsigned b = 0; b < nb; ++b) {
      std::vector<double> h(x.blocks()[b].size());
      std::mt19937_64 rng(977 + b);
      for (double &v : h) v = double(rng() >> 11) * (2.0 / 9007199254740992.0) - 1.0;
      x.blocks()[b].copy_from_host(h);
      const unsigned long long n = h.size();
      std::fwrite(&n, sizeof n, 1, f);
      std::fwrite(h.data(), sizeof(double), n, f);
    }
    gmg.vmult(y, x);
    for (unsigned b = 0; b < nb; ++b) {
      const auto h = y.blocks()[b].copy_to_host();
      std::fwrite(h.data(), sizeof(double), h.size(), f);
    }
    std::fclose(f);
    std::printf("blocks=%llu\n", nb);
    return 0;
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
}
