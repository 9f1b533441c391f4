// Prints host-side tables of the mirror for the CPU tests (tests/test_stokes_tables.py), which compare
// them with the reference's own goldens (tests/tp_02.output "Stokes ..." sections, tests/tp04.output):
//   print_tables stokes                 every (type, r, n_timesteps) of the golden's Stokes sections:
//                                       header line, then Alpha, Beta, Gamma, Zeta with 17 digits
//   print_tables blockslice             BlockSlice index / decompose / get_variable tables, one line per
//                                       entry in the wording of the golden
#include "stfem/stokes.h"

#include <cstdio>
#include <cstring>
#include <vector>

using namespace stfem;

static void print(const FullMatrix<double> &m)
{
  std::printf("%u %u\n", m.m(), m.n());
  for (unsigned i = 0; i < m.m(); ++i) {
    for (unsigned j = 0; j < m.n(); ++j) std::printf("%.17g ", m(i, j));
    std::printf("\n");
  }
}

static void stokes_section(TimeStepType type, unsigned r, unsigned ns)
{
  std::printf("Stokes %s(%u) - %u timesteps in one system\n", type == TimeStepType::CGP ? "CG" : "DG", r, ns);
  const auto w = get_fe_time_weights_stokes<double>(type, r, 1.0, ns);
  for (const auto &m : w) print(m);
}

static void blockslice_table(bool variable_major, unsigned nts, unsigned nv, unsigned ntd)
{
  BlockSlice s(nts, nv, ntd, variable_major);
  std::printf("Testing %s layout\n", variable_major ? "variable-major" : "timedof-major");
  for (unsigned it = 0; it < nts; ++it)
    for (unsigned v = 0; v < nv; ++v)
      for (unsigned d = 0; d < ntd; ++d) {
        const unsigned i = s.index(it, v, d);
        const auto t = s.decompose(i);
        std::printf("Computed Index: %u Decomposed: Timestep: %u, variable: %u, timedof: %u %s\n", i, t[0], t[1], t[2],
                    (t[0] == it && t[1] == v && t[2] == d) ? "[PASS]" : "[FAIL]");
      }
  for (unsigned it = 0; it < nts; ++it)
    for (unsigned d = 0; d < ntd; ++d) {
      // the blocks of all variables at one (timestep, timedof): variable-major they are ntd apart
      bool ok = true;
      for (unsigned v = 0; v < nv; ++v) ok = ok && s.index(it, v, d) == d + it * ntd * nv + v * ntd;
      std::printf("get_variable:  %s\n", ok ? "[PASS]" : "[FAIL]");
    }
}

int main(int argc, char **argv)
{
  if (argc != 2) return 2;
  if (!std::strcmp(argv[1], "stokes")) {
    for (unsigned r = 0; r <= 4; ++r) {
      stokes_section(TimeStepType::DG, r, 1);
      if (r < 4) stokes_section(TimeStepType::CGP, r + 1, 1);
    }
    for (unsigned ns : {1u, 2u, 4u}) {
      stokes_section(TimeStepType::CGP, 1, ns);
      stokes_section(TimeStepType::CGP, 2, ns);
      stokes_section(TimeStepType::DG, 1, ns);
      stokes_section(TimeStepType::DG, 2, ns);
    }
    return 0;
  }
  if (!std::strcmp(argv[1], "blockslice")) {
    blockslice_table(true, 2, 3, 4);
    blockslice_table(true, 1, 1, 4);
    blockslice_table(true, 2, 1, 2);
    blockslice_table(true, 1, 1, 1);
    blockslice_table(true, 1, 1, 2);
    blockslice_table(true, 2, 2, 2);
    return 0;
  }
  return 2;
}
