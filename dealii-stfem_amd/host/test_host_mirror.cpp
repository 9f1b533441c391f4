// C++ caller written against the reference's operator interface (tests/tp_01.cc:112-168 style):
// builds K, M, the temporal matrices and a SystemMatrix, applies vmult / Tvmult / vmult_slice on the
// GPU and prints checksums that the pytest driver compares against the CPU oracle.
// Usage: test_host_mirror <degree> <ncx> <ncy> <ncz> <time type 0|1> <r> <nsteps> <out.bin> [float]
#include "stfem/operators.h"

#include <cstdio>
#include <cstdlib>
#include <random>

using namespace stfem;

template <typename Number> int run(int argc, char **argv);

int main(int argc, char **argv)
{
  if (argc >= 10 && std::string(argv[9]) == "float") return run<float>(argc, argv);
  return run<double>(argc, argv);
}

template <typename Number> int run(int argc, char **argv)
{
  if (argc < 9) {
    std::fprintf(stderr, "usage: %s degree ncx ncy ncz type r nsteps out.bin\n", argv[0]);
    return 2;
  }
  const unsigned degree = std::atoi(argv[1]);
  Mesh mesh;
  for (int d = 0; d < 3; ++d) mesh.ncell[d] = std::atoi(argv[2 + d]);
  const auto type = std::atoi(argv[5]) == 0 ? TimeStepType::CGP : TimeStepType::DG;
  const unsigned r = std::atoi(argv[6]), nsteps = std::atoi(argv[7]);
  try {
    MatrixFreeOperatorScalar<3, Number> K_mf(mesh, degree, 0.0, 1.0);
    MatrixFreeOperatorScalar<3, Number> M_mf(K_mf, 1.0, 0.0);
    auto [Alpha, Beta, Gamma, Zeta] = get_fe_time_weights<Number>(type, r, 1.0 / 32, nsteps);
    using SystemN = SystemMatrix<3, Number, MatrixFreeOperatorScalar<3, Number>>;
    SystemN matrix(K_mf, M_mf, Alpha, Beta);
    const bool cgp = type == TimeStepType::CGP;
    FullMatrix<Number> zero(Gamma.m(), 1);
    SystemN rhs_matrix(K_mf, M_mf, cgp ? Gamma : zero, cgp ? Zeta : Gamma); // tp_01.cc:160-166

    BlockVectorT<Number> x, y, yT, rhs;
    matrix.initialize_dof_vector(x);
    matrix.initialize_dof_vector(y);
    matrix.initialize_dof_vector(yT);
    matrix.initialize_dof_vector(rhs);
    std::vector<std::vector<double>> hx(x.n_blocks(), std::vector<double>(x.block_size()));
    for (unsigned b = 0; b < x.n_blocks(); ++b) {
      std::mt19937_64 rng(1234 + b);
      for (double &v : hx[b]) v = double(rng() >> 11) * (2.0 / 9007199254740992.0) - 1.0;
    }
    x.copy_from_host(hx);
    matrix.vmult(y, x);
    matrix.Tvmult(yT, x);
    BlockVectorT<Number> prev;
    prev.reinit(K_mf.context(), 1);
    prev.copy_from_host({hx[0]});
    rhs_matrix.vmult_slice(rhs, prev);
    rhs_matrix.vmult_slice_add(rhs, prev);

    // the nonlinear-solver face (operators.h:1953-2050): residual = rhs - A x, vmult = Jacobian apply
    PDE<3, Number, SystemN> pde;
    pde.init(matrix, y); // rhs := y = A x  ->  residual(x) = 0, residual(0.5 x) = 0.5 A x
    BlockVectorT<Number> res0, res1, half, yj;
    for (auto *v : {&res0, &res1, &half, &yj}) pde.initialize_dof_vector(*v);
    pde.residual(res0, x);
    {
      auto hh = hx;
      for (auto &b : hh)
        for (double &v : b) v *= 0.5;
      half.copy_from_host(hh);
    }
    pde.residual(res1, half, y);
    pde.vmult(yj, x);
    // diagonals (operators.h:613-637, 1035-1045, 1106-1109)
    const auto dst_diag = matrix.get_matrix_diagonal();
    const auto dst_inv = matrix.get_matrix_diagonal_inverse();
    BlockVectorT<Number> dk, dki;
    dk.reinit(K_mf.context(), 1);
    dki.reinit(K_mf.context(), 1);
    {
      const auto d = K_mf.get_matrix_diagonal(), di = K_mf.get_matrix_diagonal_inverse();
      dk.copy_from_host({d.copy_to_host()});
      dki.copy_from_host({di.copy_to_host()});
    }

    // the smoother (stmg.h:619-907): one sweep of the cell-patch Vanka on x
    BlockVectorT<Number> sm;
    matrix.initialize_dof_vector(sm);
    {
      PreconditionVanka<Number> vanka(K_mf, Alpha, Beta);
      vanka.smooth(sm, x);
    }

    // error behaviour: aliasing and shape mismatch must throw
    int thrown = 0;
    try { matrix.vmult(x, x); } catch (const Error &e) { thrown += e.status == STFEM_ERR_ALIAS; }
    try { matrix.vmult(prev, x); } catch (const Error &e) { thrown += e.status == STFEM_ERR_SHAPE_MISMATCH; }
    try { (void)matrix.el(0, 0); } catch (const std::logic_error &) { ++thrown; }

    FILE *f = std::fopen(argv[8], "wb");
    if (!f) return 3;
    const unsigned long long nb = x.n_blocks(), n = x.block_size();
    std::fwrite(&nb, sizeof nb, 1, f);
    std::fwrite(&n, sizeof n, 1, f);
    for (const auto *v : {&hx})
      for (const auto &b : *v) std::fwrite(b.data(), sizeof(double), n, f);
    for (const auto &vec : {y.copy_to_host(), yT.copy_to_host(), rhs.copy_to_host(), res0.copy_to_host(), res1.copy_to_host(),
                            yj.copy_to_host(), dst_diag.copy_to_host(), dst_inv.copy_to_host()})
      for (const auto &b : vec) std::fwrite(b.data(), sizeof(double), n, f);
    for (const auto &vec : {dk.copy_to_host(), dki.copy_to_host()})
      for (const auto &b : vec) std::fwrite(b.data(), sizeof(double), n, f);
    for (const auto &b : sm.copy_to_host()) std::fwrite(b.data(), sizeof(double), n, f);
    std::fclose(f);
    std::printf("m=%llu blocks=%llu exceptions=%d\n", matrix.m(), nb, thrown);
    return thrown == 3 ? 0 : 4;
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
}
