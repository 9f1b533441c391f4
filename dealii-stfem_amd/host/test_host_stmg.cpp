// One V-cycle of the space-time multigrid mirror (host/stfem/stmg.h) on a small mesh, written out for the comparison with
// the numpy restatement (tests/test_gpu_stmg.py): levels as tests/tp_01.cc:170-200 derives them.
// Usage: test_host_stmg <type 0 = cG | 1 = dG> <k> <cells per direction> <n_timesteps_at_once> <fe_degree> <coarsening 0 = space_or_time |
//                       1 = space_and_time> <pmg 0|1> <double|float> <distort> <out.bin> [relaxation = 0 (estimated)] [variable = 1] [smoother 1 = relaxation | 2 = Chebyshev] [smoothing steps = 1] [coarse GMRES iterations = 0 (the smoother)]
// out.bin: uint64 {n_levels, n_blocks, n_dofs}, double relaxation[n_levels], smoother id[n_levels], src[n_blocks][n_dofs], dst[n_blocks][n_dofs]
#include "stfem/stmg.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace stfem;

template <typename NP> int run(int argc, char **argv)
{
  const auto type = std::atoi(argv[1]) == 0 ? TimeStepType::CGP : TimeStepType::DG;
  const unsigned k = std::atoi(argv[2]), nsteps = std::atoi(argv[4]), fe_degree = std::atoi(argv[5]);
  const int n = std::atoi(argv[3]);
  const auto ctype = std::atoi(argv[6]) ? CoarseningType::space_and_time : CoarseningType::space_or_time;
  const bool use_pmg = std::atoi(argv[7]) != 0;
  const double distort = std::atof(argv[9]), tau = 0.0625;
  Mesh mesh;
  mesh.ncell[0] = mesh.ncell[1] = mesh.ncell[2] = n;
  if (distort > 0) mesh.distort_random(distort, 77);
  unsigned n_sp_lvl = 1;
  for (int c = n; c % 2 == 0; c /= 2) ++n_sp_lvl;
  const auto poly_time = get_poly_mg_sequence(k, std::min(k, 1u), PolynomialCoarseningSequenceType::bisect);
  std::vector<unsigned> poly_space;
  for (unsigned q : poly_time) poly_space.push_back(q + (fe_degree - k));
  const auto mg_type_level = get_mg_sequence(n_sp_lvl, poly_time, poly_space, nsteps, 1, MGType::tau, ctype, false, use_pmg, true);
  PreconditionerGMGAdditionalData mg_data;
  if (argc > 11) mg_data.relaxation = std::atof(argv[11]);
  if (argc > 12) mg_data.variable = std::atoi(argv[12]) != 0;
  if (argc > 13 && std::atoi(argv[13]) == 2) mg_data.smoother = SupportedSmoothers::Chebyshev; // then "relaxation" in the output is the eigenvalue estimate
  if (argc > 14) mg_data.smoothing_steps = std::atoi(argv[14]);
  if (argc > 15 && std::atoi(argv[15]) > 0) { // coarseGridSmootherType != "Smoother": GMRES on the coarsest level (stmg.h:1240-1308)
    mg_data.coarse_grid_smoother_type = "Solver";
    mg_data.coarse_grid_maxiter = std::atoi(argv[15]);
  }
  STMGHierarchy<3, NP> mg(mesh, fe_degree, poly_space, type, tau, nsteps, mg_type_level, poly_time, mg_data, ctype, false, true);
  std::printf("levels:");
  for (auto m : mg_type_level) std::printf(" %c", char(m));
  std::printf("\n");

  BlockVectorT<NP> x, src, dst;
  const auto &A = *mg.operators.back();
  A.initialize_dof_vector(x);
  A.initialize_dof_vector(src);
  A.initialize_dof_vector(dst);
  const size_t N = x.block_size();
  std::vector<std::vector<double>> hx(x.n_blocks(), std::vector<double>(N));
  uint64_t state = 12345;
  for (auto &b : hx)
    for (double &v : b) {
      state = state * 6364136223846793005ull + 1442695040888963407ull;
      v = double(state >> 11) / double(1ull << 53) * 2.0 - 1.0;
    }
  x.copy_from_host(hx);
  A.vmult(src, x); // a right-hand side with zero constrained rows, like every vector FGMRES hands over
  mg.gmg->vmult(dst, src); // plain launches
  // the second application records the cycle into a hipGraph, the third replays it: all three must agree
  BlockVectorT<NP> dst2, dst3;
  A.initialize_dof_vector(dst2);
  A.initialize_dof_vector(dst3);
  mg.gmg->vmult(dst2, src);
  mg.gmg->vmult(dst3, src);
  axpby(-1.0, dst, 1.0, dst2);
  axpby(-1.0, dst, 1.0, dst3);
  std::printf("graph: recorded %.3e replayed %.3e of %.3e\n", norm(dst2), norm(dst3), norm(dst));

  FILE *f = std::fopen(argv[10], "wb");
  if (!f) return 3;
  const uint64_t head[3] = {mg.operators.size(), x.n_blocks(), N};
  std::fwrite(head, sizeof(uint64_t), 3, f);
  for (unsigned l = 0; l < head[0]; ++l) {
    const double om = mg.gmg->relaxation(l);
    std::fwrite(&om, sizeof(double), 1, f);
  }
  for (unsigned l = 0; l < head[0]; ++l) {
    const double id = mg.gmg->smoother_types()[l];
    std::fwrite(&id, sizeof(double), 1, f);
  }
  for (const auto &v : {src.copy_to_host(), dst.copy_to_host()})
    for (const auto &b : v) std::fwrite(b.data(), sizeof(double), N, f);
  std::fclose(f);
  return 0;
}

int main(int argc, char **argv)
{
  if (argc < 11) {
    std::fprintf(stderr, "usage: %s type k cells nsteps fe_degree coarsening pmg double|float distort out.bin [relaxation] [variable]\n", argv[0]);
    return 2;
  }
  try {
    return std::strcmp(argv[8], "float") == 0 ? run<float>(argc, argv) : run<double>(argc, argv);
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
}
