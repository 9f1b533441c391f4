// C++ caller of the partitioned operator (host/stfem/operators.h: set_partition, the reducing dot):
// no Python and no torch between the operator and RCCL.  One GPU can hold only one rank of an RCCL
// communicator, so this program runs the one-rank communicator with the rank itself as lower and
// upper neighbour: the ghost update copies the bottom plane of src into its top plane, and the
// add-exchange returns every interface partial to its sender (both interface planes double).  The
// pytest driver rebuilds exactly that from the CPU oracle.
// Usage: test_host_sharded <degree> <ncx> <ncy> <ncz> <out.bin> [float]
#include "stfem/operators.h"

#include <cstdio>
#include <cstdlib>
#include <random>

using namespace stfem;

template <typename Number> int run(char **argv)
{
  const unsigned degree = std::atoi(argv[1]);
  Mesh mesh;
  for (int d = 0; d < 3; ++d) mesh.ncell[d] = std::atoi(argv[2 + d]);
  mesh.dirichlet_mask = 63 & ~(16 | 32); // both z faces are partition interfaces
  try {
    auto comm = std::make_shared<Communicator>(Communicator::unique_id(), 0, 1, mesh.device);
    MatrixFreeOperatorScalar<3, Number> K_mf(mesh, degree, 0.0, 1.0);
    MatrixFreeOperatorScalar<3, Number> M_mf(K_mf, 1.0, 0.0);
    K_mf.set_partition(comm, 0, 0);
    auto [Alpha, Beta, Gamma, Zeta] = get_fe_time_weights<Number>(TimeStepType::CGP, 2, 1.0 / 32, 1);
    (void)Gamma; (void)Zeta;
    SystemMatrix<3, Number, MatrixFreeOperatorScalar<3, Number>> matrix(K_mf, M_mf, Alpha, Beta);
    BlockVectorT<Number> x, y, z;
    matrix.initialize_dof_vector(x);
    matrix.initialize_dof_vector(y);
    matrix.initialize_dof_vector(z);
    std::vector<std::vector<double>> hx(x.n_blocks(), std::vector<double>(x.block_size()));
    for (unsigned b = 0; b < x.n_blocks(); ++b) {
      std::mt19937_64 rng(77 + b);
      for (double &v : hx[b]) v = double(rng() >> 11) * (2.0 / 9007199254740992.0) - 1.0;
    }
    x.copy_from_host(hx);
    matrix.vmult(y, x);
    const double xy = dot(x, y);
    // vmult_slice_add on a partitioned context: z = y (zero + product), then z += product again
    matrix.vmult_slice(z, x);
    matrix.vmult_slice_add(z, x);
    const auto gx = x.copy_to_host(), gy = y.copy_to_host(), gz = z.copy_to_host();
    FILE *f = std::fopen(argv[5], "wb");
    if (!f) return 3;
    const unsigned long long nb = x.n_blocks(), n = x.block_size();
    std::fwrite(&nb, 8, 1, f);
    std::fwrite(&n, 8, 1, f);
    std::fwrite(&xy, 8, 1, f);
    const std::vector<std::vector<double>> *all[4] = {&hx, &gx, &gy, &gz};
    for (const auto *v : all)
      for (const auto &b : *v) std::fwrite(b.data(), 8, b.size(), f);
    std::fclose(f);
    std::printf("ranks=%d owned=%lld\n", comm->size(), (long long)K_mf.context()->n_owned());
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}

int main(int argc, char **argv)
{
  if (argc < 6) {
    std::fprintf(stderr, "usage: %s degree ncx ncy ncz out.bin [float]\n", argv[0]);
    return 2;
  }
  if (argc >= 7 && std::string(argv[6]) == "float") return run<float>(argv);
  return run<double>(argv);
}
