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

// fe_time.h:749-898: time-multigrid transfer matrices
template <typename Number = double, typename F, typename... A> FullMatrix<Number> time_transfer_matrix(F fn, const char *what, A... args)
{
  int32_t dims[2];
  if (fn(args..., nullptr, dims) != STFEM_OK) throw Error(STFEM_ERR_INVALID_ARGUMENT, what);
  FullMatrix<double> d{unsigned(dims[0]), unsigned(dims[1])};
  if (fn(args..., d.data(), dims) != STFEM_OK) throw Error(STFEM_ERR_INVALID_ARGUMENT, what);
  return d.template cast<Number>();
}
template <typename Number = double> FullMatrix<Number> get_time_prolongation_matrix(TimeStepType type, unsigned r, unsigned n_timesteps_at_once = 2)
{
  return time_transfer_matrix<Number>(stfem_time_prolongation_matrix, "get_time_prolongation_matrix", int(type), int(r), int(n_timesteps_at_once));
}
template <typename Number = double> FullMatrix<Number> get_time_restriction_matrix(TimeStepType type, unsigned r, unsigned n_timesteps_at_once = 2)
{
  return time_transfer_matrix<Number>(stfem_time_restriction_matrix, "get_time_restriction_matrix", int(type), int(r), int(n_timesteps_at_once));
}
template <typename Number = double>
FullMatrix<Number> get_time_projection_matrix(TimeStepType type, unsigned r_src, unsigned r_dst, unsigned n_timesteps_at_once)
{
  return time_transfer_matrix<Number>(stfem_time_projection_matrix, "get_time_projection_matrix", int(type), int(r_src), int(r_dst),
                                      int(n_timesteps_at_once));
}

} // namespace stfem
