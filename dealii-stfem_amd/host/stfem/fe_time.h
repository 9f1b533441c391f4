// Host-side mirror of the reference's include/fe_time.h entry points used by the hot path.
#pragma once
#include "types.h"

#include <array>

namespace stfem {

enum class TimeStepType { CGP = 0, DG = 1 }; // include/fe_time.h (TimeStepType)

// get_fe_time_weights (include/fe_time.h:351-409): {Alpha, Beta, Gamma, Zeta}
template <typename Number>
std::array<FullMatrix<Number>, 4> get_fe_time_weights(TimeStepType type, unsigned r, double time_step_size,
                                                      unsigned n_timesteps_at_once = 1)
{
  const unsigned nt = type == TimeStepType::CGP ? r : r + 1, nb = nt * n_timesteps_at_once;
  std::array<FullMatrix<double>, 4> d{{FullMatrix<double>(nb, nb), FullMatrix<double>(nb, nb),
                                       FullMatrix<double>(nb, 1), FullMatrix<double>(nb, 1)}};
  const int rc = stfem_fe_time_weights(int(type), int(r), time_step_size, int(n_timesteps_at_once),
                                       d[0].data(), d[1].data(), d[2].data(), d[3].data());
  if (rc != int(nb)) throw Error(rc, "stfem_fe_time_weights");
  std::array<FullMatrix<Number>, 4> w;
  for (int k = 0; k < 4; ++k) w[k] = d[k].template cast<Number>();
  return w;
}

// get_fe_time_weights_wave (include/fe_time.h:157-305): {Alpha_lhs, Beta_lhs, rhs_uK, rhs_uM, rhs_vM}
template <typename Number>
std::array<FullMatrix<Number>, 5> get_fe_time_weights_wave(TimeStepType type, unsigned r, double time_step_size,
                                                           unsigned n_timesteps_at_once = 1)
{
  const unsigned nt = type == TimeStepType::CGP ? r : r + 1, nb = nt * n_timesteps_at_once;
  std::array<FullMatrix<double>, 5> d{{FullMatrix<double>(nb, nb), FullMatrix<double>(nb, nb),
                                       FullMatrix<double>(nb, 1), FullMatrix<double>(nb, 1),
                                       FullMatrix<double>(nb, 1)}};
  const int rc = stfem_fe_time_weights_wave(int(type), int(r), time_step_size, int(n_timesteps_at_once),
                                            d[0].data(), d[1].data(), d[2].data(), d[3].data(), d[4].data());
  if (rc != int(nb)) throw Error(rc, "stfem_fe_time_weights_wave");
  std::array<FullMatrix<Number>, 5> w;
  for (int k = 0; k < 5; ++k) w[k] = d[k].template cast<Number>();
  return w;
}

} // namespace stfem
