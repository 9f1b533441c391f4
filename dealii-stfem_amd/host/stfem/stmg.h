// Host-side mirror of the reference's space-time multigrid (include/stmg.h:38-617 transfers,
// 968-1419 PreconditionSTMG / GMG; include/fe_time.cc:17-150 and fe_time.h:412-443 level schedule) on the C-ABI.
// The reference assembles it from deal.II parts (MGTwoLevelTransfer, Multigrid, MGSmootherPrecondition,
// PreconditionRelaxation, MGCoarseGridApplySmoother, PreconditionMG); their control flow is restated here, every
// vector operation runs on the device: level operators = the fused space-time sweep, smoother = the Vanka apply,
// space transfers = stfem_transfer_* (three banded 1D passes), time transfers = stfem_tensorproduct_add.
#pragma once
#include "stokes.h"
#include "time_integrators.h"

#include <algorithm>
#include <memory>
#include <variant>

namespace stfem {

enum class MGType : char { tau = 't', k = 'k', h = 'h', p = 'p' };                      // fe_time.h:29-35
enum class CoarseningType : int { space_or_time = 0, space_and_time = 1 };              // types.h:102-106
enum class SupportedSmoothers : unsigned { Identity = 0, Relaxation = 1, Chebyshev = 2 }; // types.h:108-113
enum class PolynomialCoarseningSequenceType { bisect = 0, decrease_by_one = 1, go_to_one = 2 };
inline bool is_space_lvl(MGType mg) { return mg == MGType::h || mg == MGType::p; }
inline bool is_time_lvl(MGType mg) { return mg == MGType::tau || mg == MGType::k; }

// fe_time.cc:40-56
inline std::vector<unsigned> get_poly_mg_sequence(unsigned k_max, unsigned k_min, PolynomialCoarseningSequenceType p_seq)
{
  int32_t n = 0;
  check(stfem_poly_mg_sequence(int(k_max), int(k_min), int(p_seq), nullptr, &n), "get_poly_mg_sequence");
  std::vector<int32_t> s(n);
  check(stfem_poly_mg_sequence(int(k_max), int(k_min), int(p_seq), s.data(), &n), "get_poly_mg_sequence");
  return std::vector<unsigned>(s.begin(), s.end());
}

// fe_time.cc:58-124
inline std::vector<MGType> get_mg_sequence(unsigned n_sp_lvl, const std::vector<unsigned> &k_seq, const std::vector<unsigned> &p_seq,
                                           unsigned n_timesteps_at_once, unsigned n_timesteps_at_once_min = 1, MGType lower_lvl = MGType::k,
                                           CoarseningType coarsening_type = CoarseningType::space_and_time, bool time_before_space = false,
                                           bool use_p_multigrid_space = false, bool zip_from_back = true)
{
  auto call = [&](char *out, int32_t *n) {
    return stfem_mg_sequence(int(n_sp_lvl), int(k_seq.size()), int(p_seq.size()), int(n_timesteps_at_once), int(n_timesteps_at_once_min),
                             char(lower_lvl), int(coarsening_type), time_before_space, use_p_multigrid_space, zip_from_back, out, n);
  };
  int32_t n = 0;
  check(call(nullptr, &n), "get_mg_sequence");
  std::string s(size_t(n), ' ');
  check(call(s.data(), &n), "get_mg_sequence");
  std::vector<MGType> r;
  for (char c : s) r.push_back(MGType(c));
  return r;
}

// fe_time.cc:126-150
inline std::vector<unsigned> get_precondition_stmg_types(const std::vector<MGType> &mg_type_level, CoarseningType coarsening_type,
                                                         bool time_before_space, bool /*zip_from_back*/,
                                                         SupportedSmoothers smoother = SupportedSmoothers::Relaxation)
{
  std::string s;
  for (MGType m : mg_type_level) s.push_back(char(m));
  std::vector<int32_t> out(s.size() + 1);
  check(stfem_precondition_stmg_types(s.data(), int(s.size()), int(coarsening_type), time_before_space, int(smoother), out.data()),
        "get_precondition_stmg_types");
  return std::vector<unsigned>(out.begin(), out.end());
}

// stmg.h:460-501: block structure of every level, finest last
inline std::vector<BlockSlice> get_blk_indices(TimeStepType type, unsigned n_timesteps_at_once, unsigned n_variables, unsigned n_levels,
                                               const std::vector<MGType> &mg_type_level, const std::vector<unsigned> &poly_time_sequence)
{
  if (mg_type_level.size() + 1 != n_levels) throw std::invalid_argument("get_blk_indices: n_levels - 1 transfers expected");
  std::vector<BlockSlice> blk(n_levels);
  auto p_mg = poly_time_sequence.rbegin();
  auto dofs = [&](unsigned r) { return type == TimeStepType::DG ? r + 1 : r; };
  unsigned i = n_levels - 1;
  for (auto mgt = mg_type_level.rbegin(); mgt != mg_type_level.rend(); ++mgt, --i) {
    blk[i] = BlockSlice(n_timesteps_at_once, n_variables, dofs(*p_mg));
    if (*mgt == MGType::k) ++p_mg;
    else if (*mgt == MGType::tau) n_timesteps_at_once /= 2;
  }
  if (p_mg != poly_time_sequence.rend() - 1) throw std::logic_error("get_blk_indices: degree sequence and k levels disagree");
  blk[0] = BlockSlice(n_timesteps_at_once, n_variables, dofs(*p_mg));
  return blk;
}

// fe_time.h:412-443: temporal matrices of every level, finest last
template <typename Number>
std::vector<std::array<FullMatrix<Number>, 4>> get_fe_time_weights(TimeStepType type, double time_step_size, unsigned n_timesteps_at_once,
                                                                   const std::vector<MGType> &mg_type_level,
                                                                   const std::vector<unsigned> &poly_time_sequence)
{
  std::vector<std::array<FullMatrix<Number>, 4>> tw(mg_type_level.size() + 1);
  auto t = tw.rbegin();
  auto p_mg = poly_time_sequence.rbegin();
  *t++ = get_fe_time_weights<Number>(type, *p_mg, time_step_size, n_timesteps_at_once);
  for (auto mgt = mg_type_level.rbegin(); mgt != mg_type_level.rend(); ++mgt, ++t) {
    if (*mgt == MGType::k) ++p_mg;
    else if (*mgt == MGType::tau) n_timesteps_at_once /= 2, time_step_size *= 2;
    *t = get_fe_time_weights<Number>(type, *p_mg, time_step_size, n_timesteps_at_once);
  }
  return tw;
}

// fe_time.h:445-476: the wave matrices {Alpha_lhs, Beta_lhs, rhs_uK, rhs_uM, rhs_vM} of every level, finest last
template <typename Number>
std::vector<std::array<FullMatrix<Number>, 5>> get_fe_time_weights_wave(TimeStepType type, double time_step_size, unsigned n_timesteps_at_once,
                                                                        const std::vector<MGType> &mg_type_level,
                                                                        const std::vector<unsigned> &poly_time_sequence)
{
  std::vector<std::array<FullMatrix<Number>, 5>> tw(mg_type_level.size() + 1);
  auto t = tw.rbegin();
  auto p_mg = poly_time_sequence.rbegin();
  *t++ = get_fe_time_weights_wave<Number>(type, *p_mg, time_step_size, n_timesteps_at_once);
  for (auto mgt = mg_type_level.rbegin(); mgt != mg_type_level.rend(); ++mgt, ++t) {
    if (*mgt == MGType::k) ++p_mg;
    else if (*mgt == MGType::tau) n_timesteps_at_once /= 2, time_step_size *= 2;
    *t = get_fe_time_weights_wave<Number>(type, *p_mg, time_step_size, n_timesteps_at_once);
  }
  return tw;
}

// parameters.h:11-31
struct PreconditionerGMGAdditionalData {
  double smoothing_range = 1;
  unsigned smoothing_degree = 5, smoothing_eig_cg_n_iterations = 20, smoothing_steps = 1;
  double relaxation = 0.0; // 0: estimated from the largest eigenvalue of P^-1 A (deal.II PreconditionRelaxation)
  std::string coarse_grid_smoother_type = "Smoother"; // anything else: GMRES on the coarsest level (stmg.h:1240-1308)
  unsigned coarse_grid_maxiter = 10;                  // include/parameters.h:25-27
  double coarse_grid_abstol = 1e-20, coarse_grid_reltol = 1e-4;
  SupportedSmoothers smoother = SupportedSmoothers::Relaxation;
  bool restrict_is_transpose_prolongate = true;
  bool variable = true;
};

// deal.II MGTwoLevelTransfer between the spaces of two contexts (h, p or both)
template <typename Number> class MGTwoLevelTransfer {
public:
  MGTwoLevelTransfer(const std::shared_ptr<Context> &fine, const std::shared_ptr<Context> &coarse) : fine_(fine), coarse_(coarse)
  {
    stfem_transfer *t = nullptr;
    // z-slab partition (the same ranks below / above on both levels): the restriction leaves partial sums in the coarse interface planes
    const int neighbours = (fine->lower_rank >= 0 ? 16 : 0) | (fine->upper_rank >= 0 ? 32 : 0);
    const int rc = stfem_transfer_create_partitioned(fine->h, coarse->h, neighbours, &t);
    if (rc != STFEM_OK) throw Error(rc, std::string("stfem_transfer_create: ") + stfem_transfer_last_error());
    t_.reset(t, stfem_transfer_destroy);
  }
  stfem_transfer *handle() const { return t_.get(); }

private:
  std::shared_ptr<Context> fine_, coarse_;
  std::shared_ptr<stfem_transfer> t_;
};

// stmg.h:38-110 (one variable: every block is transferred alike, all blocks in one call)
template <typename Number> class MGTwoLevelTransferSpace {
public:
  using BlockVectorType = BlockVectorT<Number>;
  MGTwoLevelTransferSpace() = default;
  MGTwoLevelTransferSpace(const BlockSlice &blk_index, const std::shared_ptr<MGTwoLevelTransfer<Number>> &transfer)
    : blk_index(blk_index), transfer(transfer)
  {}
  void prolongate_and_add(BlockVectorType &dst, const BlockVectorType &src, void *stream = nullptr) const
  {
    check(stfem_transfer_prolongate(transfer->handle(), dst.handle(), src.handle(), 1, stream), "MGTwoLevelTransferSpace::prolongate_and_add");
  }
  void restrict_and_add(BlockVectorType &dst, const BlockVectorType &src, void *stream = nullptr) const
  {
    check(stfem_transfer_restrict(transfer->handle(), dst.handle(), src.handle(), 1, stream), "MGTwoLevelTransferSpace::restrict_and_add");
    compress_add(*dst.context(), dst.handle(), stream); // partitioned: dst.compress(add) of MGTwoLevelTransfer::restrict_and_add
  }
  void interpolate(BlockVectorType &dst, const BlockVectorType &src, void *stream = nullptr) const
  {
    check(stfem_transfer_interpolate(transfer->handle(), dst.handle(), src.handle(), stream), "MGTwoLevelTransferSpace::interpolate");
  }

private:
  BlockSlice blk_index;
  std::shared_ptr<MGTwoLevelTransfer<Number>> transfer;
};

// stmg.h:113-247
template <typename Number> class MGTwoLevelTransferTime {
public:
  using BlockVectorType = BlockVectorT<Number>;
  MGTwoLevelTransferTime() = default;
  MGTwoLevelTransferTime(const BlockSlice &blk_index_hi_, const BlockSlice &blk_index_lo_, TimeStepType type,
                         bool restrict_is_transpose_prolongate, MGType mg_type)
    : blk_index_hi(blk_index_hi_), blk_index_lo(blk_index_lo_)
  {
    if ((blk_index_hi.n_timedofs() == blk_index_lo.n_timedofs()) == (blk_index_hi.n_timesteps_at_once() == blk_index_lo.n_timesteps_at_once()))
      throw std::logic_error("MGTwoLevelTransferTime: exactly one of degree and step count changes");
    if (mg_type != MGType::k && mg_type != MGType::tau) throw std::invalid_argument("MGTwoLevelTransferTime: k or tau");
    const bool k_mg = mg_type == MGType::k, dg = type == TimeStepType::DG;
    const unsigned r = dg ? blk_index_hi.n_timedofs() - 1 : blk_index_hi.n_timedofs();
    const unsigned r_lo = dg ? blk_index_lo.n_timedofs() - 1 : blk_index_lo.n_timedofs();
    const unsigned n = blk_index_hi.n_timesteps_at_once();
    prolongation_matrix = k_mg ? get_time_projection_matrix<Number>(type, r_lo, r, n) : get_time_prolongation_matrix<Number>(type, r, n);
    interpolate_down_matrix = k_mg ? get_time_projection_matrix<Number>(type, r, r_lo, n) : get_time_restriction_matrix<Number>(type, r, n);
    const FullMatrix<Number> &source = restrict_is_transpose_prolongate ? prolongation_matrix : interpolate_down_matrix;
    if (restrict_is_transpose_prolongate) {
      restriction_matrix = FullMatrix<Number>(source.n(), source.m());
      for (unsigned i = 0; i < source.m(); ++i)
        for (unsigned j = 0; j < source.n(); ++j) restriction_matrix(j, i) = source(i, j);
    } else restriction_matrix = source;
  }
  void prolongate_and_add(BlockVectorType &dst, const BlockVectorType &src, void *stream = nullptr) const
  {
    transfer_and_add(dst, blk_index_hi, prolongation_matrix, src, blk_index_lo, stream);
  }
  void restrict_and_add(BlockVectorType &dst, const BlockVectorType &src, void *stream = nullptr) const
  {
    transfer_and_add(dst, blk_index_lo, restriction_matrix, src, blk_index_hi, stream);
  }
  void interpolate(BlockVectorType &dst, const BlockVectorType &src, void *stream = nullptr) const
  {
    if (dst.n_blocks() > src.n_blocks()) throw std::invalid_argument("Interpolation only from fine to coarse");
    set_zero(dst, stream);
    transfer_and_add(dst, blk_index_lo, interpolate_down_matrix, src, blk_index_hi, stream);
  }
  const FullMatrix<Number> &prolongation() const { return prolongation_matrix; }
  const FullMatrix<Number> &restriction() const { return restriction_matrix; }

private:
  void transfer_and_add(BlockVectorType &dst, const BlockSlice &blk_dst, const FullMatrix<Number> &matrix, const BlockVectorType &src,
                        const BlockSlice &blk_src, void *stream) const
  {
    if (blk_src.n_variables() != 1) throw std::invalid_argument("MGTwoLevelTransferTime: one variable");
    if (matrix.n() != src.n_blocks() || matrix.m() != dst.n_blocks() || blk_src.n_blocks() != src.n_blocks() || blk_dst.n_blocks() != dst.n_blocks())
      throw std::invalid_argument("MGTwoLevelTransferTime: block counts");
    std::vector<double> a(size_t(matrix.m()) * matrix.n());
    for (size_t i = 0; i < a.size(); ++i) a[i] = double(matrix.data()[i]);
    check(stfem_tensorproduct_add(dst.context()->h, int(matrix.m()), int(matrix.n()), a.data(), dst.handle(), src.handle(), stream),
          "MGTwoLevelTransferTime: tensorproduct_add");
  }
  BlockSlice blk_index_hi, blk_index_lo;
  FullMatrix<Number> prolongation_matrix, restriction_matrix, interpolate_down_matrix;
};

// stmg.h:249-302
template <typename Number> class TwoLevelTransferOperator {
public:
  using BlockVectorType = BlockVectorT<Number>;
  TwoLevelTransferOperator() = default;
  TwoLevelTransferOperator(const MGTwoLevelTransferSpace<Number> &t) : transfer_variant(t) {}
  TwoLevelTransferOperator(const MGTwoLevelTransferTime<Number> &t) : transfer_variant(t) {}
  void prolongate_and_add(BlockVectorType &dst, const BlockVectorType &src, void *stream = nullptr) const
  {
    std::visit([&](auto &t) { t.prolongate_and_add(dst, src, stream); }, transfer_variant);
  }
  void restrict_and_add(BlockVectorType &dst, const BlockVectorType &src, void *stream = nullptr) const
  {
    std::visit([&](auto &t) { t.restrict_and_add(dst, src, stream); }, transfer_variant);
  }
  void interpolate(BlockVectorType &dst, const BlockVectorType &src, void *stream = nullptr) const
  {
    std::visit([&](auto &t) { t.interpolate(dst, src, stream); }, transfer_variant);
  }

private:
  std::variant<MGTwoLevelTransferSpace<Number>, MGTwoLevelTransferTime<Number>> transfer_variant;
};

// stmg.h:304-458: transfer[l] connects level l - 1 and level l
template <typename Number> class STMGTransferBlockMatrixFree {
public:
  using BlockVectorType = BlockVectorT<Number>;
  STMGTransferBlockMatrixFree(std::vector<TwoLevelTransferOperator<Number>> transfers, std::vector<BlockSlice> blk_indices,
                              std::vector<std::shared_ptr<Context>> level_contexts)
    : transfer(std::move(transfers)), blk_indices(std::move(blk_indices)), contexts(std::move(level_contexts))
  {}
  void prolongate(unsigned to_level, BlockVectorType &dst, const BlockVectorType &src, void *stream = nullptr) const
  {
    set_zero(dst, stream);
    prolongate_and_add(to_level, dst, src, stream);
  }
  void prolongate_and_add(unsigned to_level, BlockVectorType &dst, const BlockVectorType &src, void *stream = nullptr) const
  {
    transfer[to_level].prolongate_and_add(dst, src, stream);
  }
  void restrict_and_add(unsigned from_level, BlockVectorType &dst, const BlockVectorType &src, void *stream = nullptr) const
  {
    transfer[from_level].restrict_and_add(dst, src, stream);
  }
  void interpolate(unsigned from_level, BlockVectorType &dst, const BlockVectorType &src, void *stream = nullptr) const
  {
    transfer[from_level].interpolate(dst, src, stream);
  }
  // copy_to_mg (stmg.h:401-415): every level vector sized and zeroed, the finest takes src
  template <typename Number2> void copy_to_mg(std::vector<BlockVectorType> &dst, const BlockVectorT<Number2> &src, void *stream = nullptr) const
  {
    dst.resize(blk_indices.size());
    for (size_t l = 0; l < dst.size(); ++l)
      if (!dst[l].handle() || dst[l].n_blocks() != blk_indices[l].n_blocks()) dst[l].reinit(contexts[l], blk_indices[l].n_blocks());
    check(stfem_vector_convert(dst.back().handle(), src.handle(), stream), "copy_to_mg"); // copy_locally_owned_data_from
  }
  // the zeroing of the coarser level vectors (initialize_dof_vector without omit_zeroing_entries, stmg.h:408-412)
  void zero_coarser(std::vector<BlockVectorType> &dst, void *stream = nullptr) const
  {
    for (size_t l = 0; l + 1 < dst.size(); ++l) set_zero(dst[l], stream);
  }
  unsigned n_levels() const { return unsigned(blk_indices.size()); }
  const BlockSlice &blk(unsigned l) const { return blk_indices[l]; }
  const std::shared_ptr<Context> &context(unsigned l) const { return contexts[l]; }

private:
  std::vector<TwoLevelTransferOperator<Number>> transfer; // [0] unused
  std::vector<BlockSlice> blk_indices;
  std::vector<std::shared_ptr<Context>> contexts;
};

// The largest eigenvalue of P^-1 A by deal.II's power iteration (what PreconditionRelaxation does when its
// relaxation parameter is 0, as the reference leaves it: parameters.h:19, stmg.h:1207-1213): start vector
// (i mod 11) - mean on every block, n_iterations steps; relaxation = 2 / (alpha + 1.2 lambda), alpha = min(1.08 lambda, 1) for
// smoothing_range <= 1 (restated from deal.II's documentation of PreconditionRelaxation / PreconditionChebyshev; not checked against a build)
template <typename Number, typename Operator, typename Precond> double estimate_max_eigenvalue(const Operator &A, const Precond &P, unsigned n_iterations);
template <typename Number, typename Operator, typename Precond>
double estimate_relaxation(const Operator &A, const Precond &P, unsigned n_iterations, double smoothing_range)
{
  const double lambda = estimate_max_eigenvalue<Number>(A, P, n_iterations);
  if (!(lambda > 0) || !std::isfinite(lambda)) return 1.0; // a level without free DoFs (one Q1 cell, all nodes constrained): nothing to relax
  // internal::PreconditionChebyshevImplementation::estimate_eigenvalues with the power iteration: the tracker holds {1, lambda}, i.e. the
  // lower estimate is 1 and the upper one carries the safety factor 1.2
  const double beta = 1.2 * lambda, alpha = smoothing_range > 1.0 ? beta / smoothing_range : std::min(0.9 * beta, 1.0);
  return 2.0 / (alpha + beta);
}
template <typename Number, typename Operator, typename Precond> double estimate_max_eigenvalue(const Operator &A, const Precond &P, unsigned n_iterations)
{
  BlockVectorT<Number> v, w, z;
  A.initialize_dof_vector(v);
  A.initialize_dof_vector(w);
  A.initialize_dof_vector(z);
  const size_t n = v.block_size();
  std::vector<double> guess(n);
  double mean = 0.0;
  for (size_t i = 0; i < n; ++i) mean += double(i % 11);
  mean /= double(n);
  for (size_t i = 0; i < n; ++i) guess[i] = double(i % 11) - mean;
  v.copy_from_host(std::vector<std::vector<double>>(v.n_blocks(), guess));
  axpby(0.0, v, 1.0 / norm(v), v);
  double lambda = 0.0;
  for (unsigned it = 0; it < n_iterations; ++it) {
    A.vmult(z, v);
    P.vmult(w, z);
    lambda = dot(v, w);
    const double nw = norm(w);
    if (!(nw > 0)) break;
    axpby(1.0 / nw, w, 0.0, v);
  }
  return std::abs(lambda);
}

// deal.II PreconditionChebyshev<LevelMatrix, BlockVector, PreconditionVanka> as the reference's second smoother alternative
// sets it up (stmg.h:1216-1227: degree = smoothing_steps, power iteration for the largest eigenvalue of P^-1 A, smoothing_range):
// Chebyshev iteration on [alpha, 1.2 lambda] from x = 0, `degree` applications of P^-1 (the first without a matrix product):
//   x_1 = P^-1 b / theta;   x_{k+1} = x_k + rho_{k+1} rho_k (x_k - x_{k-1}) + 2 rho_{k+1} / delta P^-1 (b - A x_k),
//   theta = (beta + alpha) / 2, delta = (beta - alpha) / 2, sigma = theta / delta, rho_1 = 1 / sigma, rho_{k+1} = 1 / (2 sigma - rho_k)
template <typename Number, typename Operator> class PreconditionChebyshev {
public:
  PreconditionChebyshev(const Operator &A, const PreconditionVanka<Number> &P, double lambda_max, double smoothing_range, unsigned degree)
    : A(A), P(P), degree(degree)
  {
    const double beta = 1.2 * lambda_max; // deal.II's safety factor on the estimate; the lower estimate of the power iteration is 1
    const double alpha = smoothing_range > 1.0 ? beta / smoothing_range : std::min(0.9 * beta, 1.0);
    theta = 0.5 * (beta + alpha);
    delta = 0.5 * (beta - alpha);
  }
  void vmult(BlockVectorT<Number> &dst, const BlockVectorT<Number> &src, void *stream = nullptr) const
  {
    if (!tmp.handle()) {
      A.initialize_dof_vector(tmp);
      A.initialize_dof_vector(res);
      A.initialize_dof_vector(old);
    }
    P.vmult(dst, src, stream);
    axpby(0.0, dst, 1.0 / theta, dst, stream); // x_1
    if (degree < 2) return;
    set_zero(old, stream);                      // x_0 = 0
    const double sigma = theta / delta;
    double rho = 1.0 / sigma;
    for (unsigned k = 1; k < degree; ++k) {
      const double rho_new = 1.0 / (2.0 * sigma - rho), f1 = rho_new * rho, f2 = 2.0 * rho_new / delta;
      A.vmult(res, dst, stream);
      axpby(1.0, src, -1.0, res, stream); // res = b - A x_k
      P.vmult(tmp, res, stream);
      // old <- x_k + f1 (x_k - old) + f2 tmp, then swap the roles of old and dst
      axpby(1.0 + f1, dst, -f1, old, stream);
      axpby(f2, tmp, 1.0, old, stream);
      axpby(1.0, dst, 0.0, tmp, stream); // tmp = x_k
      axpby(1.0, old, 0.0, dst, stream); // dst = x_{k+1}
      axpby(1.0, tmp, 0.0, old, stream); // old = x_k
      rho = rho_new;
    }
  }

private:
  const Operator &A;
  const PreconditionVanka<Number> &P;
  unsigned degree;
  double theta = 1.0, delta = 1.0;
  mutable BlockVectorT<Number> tmp, res, old;
};

// stmg.h:968-1045 PreconditionSTMG: the smoother of one level - identity, or relaxation sweeps of the Vanka smoother
// or the Chebyshev iteration around it
template <typename Number, typename LevelMatrixType> class PreconditionSTMG {
public:
  using BlockVectorType = BlockVectorT<Number>;
  PreconditionSTMG() = default;
  void initialize(const LevelMatrixType &matrix, const std::shared_ptr<PreconditionVanka<Number>> &vanka, double relaxation, unsigned n_iterations)
  {
    relax = std::make_unique<PreconditionRelaxation<Number, LevelMatrixType>>(matrix, *vanka, relaxation, n_iterations);
    keep = vanka;
    omega = relaxation;
  }
  void initialize_chebyshev(const LevelMatrixType &matrix, const std::shared_ptr<PreconditionVanka<Number>> &vanka, double lambda_max, double smoothing_range,
                            unsigned degree)
  {
    cheb = std::make_unique<PreconditionChebyshev<Number, LevelMatrixType>>(matrix, *vanka, lambda_max, smoothing_range, degree);
    keep = vanka;
    omega = lambda_max;
  }
  void vmult(BlockVectorType &dst, const BlockVectorType &src, void *stream = nullptr) const
  {
    if (relax) relax->vmult(dst, src, stream);
    else if (cheb) cheb->vmult(dst, src, stream);
    else axpby(1.0, src, 0.0, dst, stream);
  }
  void smooth(BlockVectorType &u, const BlockVectorType &rhs) const { vmult(u, rhs); }
  bool is_identity() const { return !relax && !cheb; }
  double relaxation() const { return omega; }

private:
  std::unique_ptr<PreconditionRelaxation<Number, LevelMatrixType>> relax;
  std::unique_ptr<PreconditionChebyshev<Number, LevelMatrixType>> cheb;
  std::shared_ptr<PreconditionVanka<Number>> keep;
  double omega = 1.0;
};

// stmg.h:1047-1419 GMG: the V-cycle as FGMRES preconditioner.  Restates what the reference assembles from deal.II:
// PreconditionMG::vmult (copy_to_mg, cycle, copy_from_mg), Multigrid::level_v_step, MGSmootherPrecondition
// (steps = 1, variable: 2^(max_level - level) steps on level `level`) and MGCoarseGridApplySmoother.
template <int dim, typename Number, typename LevelMatrixType> class GMG {
public:
  using BlockVectorType = BlockVectorT<Number>;
  GMG(const PreconditionerGMGAdditionalData &additional_data, TimeStepType type, unsigned n_timesteps_at_once,
      const std::vector<MGType> &mg_type_level, const std::vector<unsigned> &poly_time_sequence, CoarseningType coarsening_type,
      bool time_before_space, bool space_time_level_first, const std::vector<std::shared_ptr<const LevelMatrixType>> &mg_operators,
      const std::vector<std::shared_ptr<PreconditionVanka<Number>>> &mg_smoother_)
    : additional_data(additional_data), mg_sequence(mg_type_level),
      precondition_sequence(get_precondition_stmg_types(mg_type_level, coarsening_type, time_before_space, space_time_level_first, additional_data.smoother)),
      mg_operators(mg_operators), precondition_vanka(mg_smoother_)
  {
    const unsigned n_levels = unsigned(mg_operators.size());
    if (additional_data.coarse_grid_smoother_type != "Smoother") use_graph = false; // the coarse GMRES reads its inner products back every step
    std::vector<BlockSlice> blk_indices = get_blk_indices(type, n_timesteps_at_once, 1u, n_levels, mg_type_level, poly_time_sequence);
    // build_stmg_transfers (stmg.h:503-617)
    std::vector<std::shared_ptr<Context>> contexts(n_levels);
    for (unsigned l = 0; l < n_levels; ++l) {
      BlockVectorType probe;
      mg_operators[l]->initialize_dof_vector(probe);
      contexts[l] = probe.context();
      if (probe.n_blocks() != blk_indices[l].n_blocks()) throw std::invalid_argument("GMG: level operator and block structure disagree");
    }
    std::vector<TwoLevelTransferOperator<Number>> transfers(n_levels);
    for (unsigned l = n_levels - 1; l >= 1; --l) {
      const MGType mgt = mg_type_level[l - 1];
      if (is_space_lvl(mgt))
        transfers[l] = MGTwoLevelTransferSpace<Number>(blk_indices[l], std::make_shared<MGTwoLevelTransfer<Number>>(contexts[l], contexts[l - 1]));
      else
        transfers[l] = MGTwoLevelTransferTime<Number>(blk_indices[l], blk_indices[l - 1], type, additional_data.restrict_is_transpose_prolongate, mgt);
    }
    transfer_block = std::make_unique<STMGTransferBlockMatrixFree<Number>>(std::move(transfers), std::move(blk_indices), std::move(contexts));
  }

  // stmg.h:1190-1327 (the Relaxation / "Smoother" branches)
  void reinit()
  {
    const unsigned n_levels = unsigned(mg_operators.size());
    mg_smoother.clear();
    mg_smoother.resize(n_levels);
    for (unsigned l = 0; l < n_levels; ++l) {
      if (precondition_sequence[l] == unsigned(SupportedSmoothers::Identity)) continue;
      if (precondition_sequence[l] == unsigned(SupportedSmoothers::Chebyshev)) { // stmg.h:1216-1227
        const double lambda = estimate_max_eigenvalue<Number>(*mg_operators[l], *precondition_vanka[l], additional_data.smoothing_eig_cg_n_iterations);
        mg_smoother[l].initialize_chebyshev(*mg_operators[l], precondition_vanka[l], lambda > 0 && std::isfinite(lambda) ? lambda : 1.0,
                                            additional_data.smoothing_range, additional_data.smoothing_steps);
        continue;
      }
      double omega = additional_data.relaxation;
      if (omega == 0.0)
        omega = estimate_relaxation<Number>(*mg_operators[l], *precondition_vanka[l], additional_data.smoothing_eig_cg_n_iterations,
                                            additional_data.smoothing_range);
      mg_smoother[l].initialize(*mg_operators[l], precondition_vanka[l], omega, additional_data.smoothing_steps);
    }
    defect.clear();
    solution.clear();
    t.clear();
    d_.clear();
    graph_.reset();
    cycles_run = 0;
  }

  // PreconditionMG::vmult.  Vectors of another Number (the solver's double against the multigrid's float, stmg.h:1330-1343)
  // or of another context on the same mesh are converted on the way into and out of the level vectors.
  template <typename Number2> void vmult(BlockVectorT<Number2> &dst, const BlockVectorT<Number2> &src) const { cycle(dst, src); }
  void interpolate(unsigned level, BlockVectorType &dst, const BlockVectorType &src) const { transfer_block->interpolate(level, dst, src); }
  const STMGTransferBlockMatrixFree<Number> &transfer() const { return *transfer_block; }
  double relaxation(unsigned level) const { return mg_smoother[level].relaxation(); }
  const std::vector<unsigned> &smoother_types() const { return precondition_sequence; }

private:
  unsigned steps_on(unsigned level) const
  {
    const unsigned max_level = unsigned(mg_operators.size()) - 1;
    return additional_data.variable ? 1u << (max_level - level) : 1u;
  }
  // MGSmootherPrecondition::apply (u is overwritten) / ::smooth (u is improved)
  void smooth(unsigned level, BlockVectorType &u, const BlockVectorType &rhs, bool from_zero) const
  {
    const unsigned steps = steps_on(level);
    BlockVectorType &r = t[level], &d = d_[level];
    unsigned i = 0;
    if (from_zero) {
      mg_smoother[level].vmult(u, rhs, stream_);
      i = 1;
    }
    for (; i < steps; ++i) {
      mg_operators[level]->vmult(r, u, stream_);
      axpby(1.0, rhs, -1.0, r, stream_);
      mg_smoother[level].vmult(d, r, stream_);
      axpby(1.0, d, 1.0, u, stream_);
    }
  }
  // Multigrid::level_v_step
  void level_v_step(unsigned level) const
  {
    if (level == 0) {
      if (additional_data.coarse_grid_smoother_type != "Smoother") coarse_gmres(solution[0], defect[0]); // MGCoarseGridIterativeSolver
      else smooth(0, solution[0], defect[0], true);                                                       // MGCoarseGridApplySmoother
      return;
    }
    smooth(level, solution[level], defect[level], true);
    mg_operators[level]->vmult(t[level], solution[level], stream_);
    axpby(1.0, defect[level], -1.0, t[level], stream_);
    transfer_block->restrict_and_add(level, defect[level - 1], t[level], stream_);
    level_v_step(level - 1);
    transfer_block->prolongate(level, t[level], solution[level - 1], stream_);
    axpby(1.0, t[level], 1.0, solution[level], stream_);
    smooth(level, solution[level], defect[level], false);
  }
  // The reference's coarse solver for coarseGridSmootherType != "Smoother" (stmg.h:1240-1308): SolverGMRES with
  // IterationNumberControl(coarse_grid_maxiter, coarse_grid_abstol) and a basis of coarse_grid_maxiter vectors (no restart),
  // left-preconditioned (deal.II's default) by the coarsest level's relaxation / Chebyshev preconditioner with
  // smoothing_steps sweeps, started from zero (MGCoarseGridIterativeSolver sets dst = 0).  The residual that is
  // controlled is the preconditioned one.
  void coarse_gmres(BlockVectorType &x, const BlockVectorType &b) const
  {
    const unsigned m = std::max(1u, additional_data.coarse_grid_maxiter);
    const double tol = additional_data.coarse_grid_abstol;
    set_zero(x, stream_);
    if (cv_.size() < m + 1) cv_.resize(m + 1);
    for (auto &v : cv_)
      if (!v.handle()) v.reinit(x.context(), x.n_blocks());
    if (!cw_.handle()) cw_.reinit(x.context(), x.n_blocks());
    if (precondition_sequence[0] == unsigned(SupportedSmoothers::Identity)) axpby(1.0, b, 0.0, cv_[0], stream_);
    else mg_smoother[0].vmult(cv_[0], b, stream_);
    const double beta = norm(cv_[0]);
    if (!(beta > tol)) return;
    axpby(0.0, cv_[0], 1.0 / beta, cv_[0], stream_);
    std::vector<double> H(size_t(m + 1) * m, 0.0), cs(m), sn(m), g(m + 1, 0.0), hcol(m);
    g[0] = beta;
    unsigned j = 0;
    for (; j < m;) {
      mg_operators[0]->vmult(cw_, cv_[j], stream_);
      if (precondition_sequence[0] == unsigned(SupportedSmoothers::Identity)) axpby(1.0, cw_, 0.0, cv_[j + 1], stream_);
      else mg_smoother[0].vmult(cv_[j + 1], cw_, stream_);
      const double hn = orthogonalize(cv_, j + 1, cv_[j + 1], hcol.data());
      for (unsigned i = 0; i <= j; ++i) H[i * m + j] = hcol[i];
      H[(j + 1) * m + j] = hn;
      if (hn > 0) axpby(0.0, cv_[j + 1], 1.0 / hn, cv_[j + 1], stream_);
      for (unsigned i = 0; i < j; ++i) {
        const double t1 = cs[i] * H[i * m + j] + sn[i] * H[(i + 1) * m + j];
        H[(i + 1) * m + j] = -sn[i] * H[i * m + j] + cs[i] * H[(i + 1) * m + j];
        H[i * m + j] = t1;
      }
      const double d = std::hypot(H[j * m + j], H[(j + 1) * m + j]);
      cs[j] = H[j * m + j] / d;
      sn[j] = H[(j + 1) * m + j] / d;
      H[j * m + j] = d;
      H[(j + 1) * m + j] = 0.0;
      g[j + 1] = -sn[j] * g[j];
      g[j] = cs[j] * g[j];
      ++j;
      if (std::abs(g[j]) <= tol || !(hn > 0)) break;
    }
    std::vector<double> y(j);
    for (int i = int(j) - 1; i >= 0; --i) {
      double sum = g[i];
      for (unsigned k = i + 1; k < j; ++k) sum -= H[i * m + k] * y[k];
      y[i] = sum / H[i * m + i];
    }
    for (unsigned i = 0; i < j; ++i) axpby(y[i], cv_[i], 1.0, x, stream_);
  }
  // One cycle = a fixed sequence of ~10^3 launches on the level vectors this object owns.  With STFEM_MG_GRAPH=1 the first
  // call runs it as it is (the operators allocate their scratch on first use), the second records it into a hipGraph
  // (stfem_graph_*), every later call replays the graph; only the copies between the caller's vectors and the finest level
  // vectors stay outside.  Off by default: measured on the cfg-1 mesh (profiles/r2/stmg/driver.txt) the replay takes 21.7 ms per
  // FGMRES iteration against 20.9 ms of plain launches - the small kernels of the coarse levels (>= 17 us each) are bound by
  // their own dependent-load chains, the launches already overlap them.
  template <typename Number2> void cycle(BlockVectorT<Number2> &dst, const BlockVectorT<Number2> &src) const
  {
    TraceRange scope("gmg"); // stmg.h:1335, 1352
    if (!stream_ && use_graph) check(stfem_stream_create(&stream_.s), "stfem_stream_create");
    transfer_block->copy_to_mg(defect, src, stream_);
    const unsigned n_levels = transfer_block->n_levels();
    if (solution.size() != n_levels) {
      solution.resize(n_levels);
      t.resize(n_levels);
      d_.resize(n_levels);
      for (unsigned l = 0; l < n_levels; ++l) {
        solution[l].reinit(transfer_block->context(l), transfer_block->blk(l).n_blocks());
        t[l].reinit(transfer_block->context(l), transfer_block->blk(l).n_blocks());
        d_[l].reinit(transfer_block->context(l), transfer_block->blk(l).n_blocks());
      }
    }
    auto body = [&] {
      transfer_block->zero_coarser(defect, stream_);
      level_v_step(n_levels - 1);
    };
    if (graph_) {
      check(stfem_graph_launch(graph_.get(), stream_), "stfem_graph_launch");
    } else if (use_graph && cycles_run >= 1) {
      check(stfem_graph_begin(stream_), "stfem_graph_begin");
      stfem_graph *g = nullptr;
      try {
        body();
      } catch (...) {
        (void)stfem_graph_end(stream_, &g);
        stfem_graph_destroy(g);
        throw;
      }
      const int rc = stfem_graph_end(stream_, &g);
      if (rc != STFEM_OK) throw Error(rc, std::string("stfem_graph_end: ") + stfem_transfer_last_error());
      graph_.reset(g, stfem_graph_destroy);
      check(stfem_graph_launch(graph_.get(), stream_), "stfem_graph_launch");
    } else {
      body();
    }
    ++cycles_run;
    check(stfem_vector_convert(dst.handle(), solution.back().handle(), stream_), "copy_from_mg");
  }

  PreconditionerGMGAdditionalData additional_data;
  std::vector<MGType> mg_sequence;
  std::vector<unsigned> precondition_sequence;
  std::vector<std::shared_ptr<const LevelMatrixType>> mg_operators;
  std::vector<std::shared_ptr<PreconditionVanka<Number>>> precondition_vanka;
  std::unique_ptr<STMGTransferBlockMatrixFree<Number>> transfer_block;
  std::vector<PreconditionSTMG<Number, LevelMatrixType>> mg_smoother;
  mutable std::vector<BlockVectorType> defect, solution, t, d_;
  mutable std::vector<BlockVectorType> cv_; // Krylov basis of the coarse GMRES
  mutable BlockVectorType cw_;
  // the cycle's own stream (created on the first graph capture); the holder destroys it with the object,
  // after the graph (members are destroyed in reverse order of declaration)
  struct StreamHolder {
    void *s = nullptr;
    StreamHolder() = default;
    StreamHolder(const StreamHolder &) = delete;
    StreamHolder &operator=(const StreamHolder &) = delete;
    ~StreamHolder()
    {
      if (s) stfem_stream_destroy(s);
    }
    operator void *() const { return s; }
    bool operator!() const { return s == nullptr; }
  };
  mutable StreamHolder stream_;
  mutable std::shared_ptr<stfem_graph> graph_;
  mutable unsigned cycles_run = 0;
  bool use_graph = [] {
    const char *e = std::getenv("STFEM_MG_GRAPH");
    return e && std::atoi(e) != 0;
  }();
};

// The level hierarchy as tests/tp_01.cc:170-330 sets it up: one mesh per h level (every second vertex plane of the
// finer one: global coarsening of a refined mesh), FE_Q(space degree of the level), K = (0, 1) and M = (1, 0) operators,
// the level's temporal matrices, the space-time operator and its Vanka smoother
template <int dim, typename Number> struct STMGHierarchy {
  using Operator = MatrixFreeOperatorScalar<dim, Number>;
  using System = SystemMatrix<dim, Number, Operator>;
  std::vector<MGType> mg_type_level;
  std::vector<unsigned> poly_time_sequence;
  std::vector<std::array<FullMatrix<Number>, 4>> fetw;   // heat (tests/tp_01.cc:226-233)
  std::vector<std::array<FullMatrix<Number>, 5>> fetw_w; // wave (234-241)
  std::vector<std::shared_ptr<Operator>> K, M;
  std::vector<std::shared_ptr<const System>> operators;
  std::vector<std::shared_ptr<PreconditionVanka<Number>>> vanka;
  std::unique_ptr<GMG<dim, Number, System>> gmg;

  // poly_space_sequence: spatial degree per p level, coarsest first (used if the schedule holds 'p' transfers)
  STMGHierarchy(const Mesh &fine_mesh, unsigned fe_degree_space, const std::vector<unsigned> &poly_space_sequence, TimeStepType type, double time_step_size,
                unsigned n_timesteps_at_once, const std::vector<MGType> &mg_type_level_, const std::vector<unsigned> &poly_time_sequence_,
                const PreconditionerGMGAdditionalData &mg_data, CoarseningType coarsening_type, bool time_before_space, bool space_time_level_first,
                bool wave = false, const std::function<void(Operator &, const Mesh &)> &evaluate_coefficient = {})
    : mg_type_level(mg_type_level_), poly_time_sequence(poly_time_sequence_)
  {
    const unsigned n_levels = unsigned(mg_type_level.size()) + 1;
    if (wave) fetw_w = get_fe_time_weights_wave<Number>(type, time_step_size, n_timesteps_at_once, mg_type_level, poly_time_sequence);
    else fetw = get_fe_time_weights<Number>(type, time_step_size, n_timesteps_at_once, mg_type_level, poly_time_sequence);
    K.resize(n_levels);
    M.resize(n_levels);
    operators.resize(n_levels);
    vanka.resize(n_levels);
    Mesh mesh = fine_mesh;
    unsigned degree = fe_degree_space;
    auto p_it = poly_space_sequence.rbegin();
    for (unsigned l = n_levels; l-- > 0;) {
      const bool new_space = l == n_levels - 1 || is_space_lvl(mg_type_level[l]);
      if (l < n_levels - 1 && mg_type_level[l] == MGType::h) mesh = coarsen(mesh);
      if (l < n_levels - 1 && mg_type_level[l] == MGType::p) {
        if (p_it == poly_space_sequence.rend() || ++p_it == poly_space_sequence.rend()) throw std::invalid_argument("STMGHierarchy: spatial degree sequence too short");
        degree = *p_it;
      }
      if (new_space) {
        K[l] = std::make_shared<Operator>(mesh, degree, 0.0, 1.0);
        M[l] = std::make_shared<Operator>(*K[l], 1.0, 0.0);
        if (evaluate_coefficient) evaluate_coefficient(*K[l], mesh); // K_mf_->evaluate_coefficient(coeff) on every level (tests/tp_01.cc:273-274)
      } else {
        K[l] = K[l + 1];
        M[l] = M[l + 1];
      }
      const FullMatrix<Number> &lhs_uK = wave ? fetw_w[l][0] : fetw[l][0], &lhs_uM = wave ? fetw_w[l][1] : fetw[l][1]; // tests/tp_01.cc:279-282
      operators[l] = std::make_shared<const System>(*K[l], *M[l], lhs_uK, lhs_uM);
      vanka[l] = std::make_shared<PreconditionVanka<Number>>(*K[l], lhs_uK, lhs_uM);
    }
    gmg = std::make_unique<GMG<dim, Number, System>>(mg_data, type, n_timesteps_at_once, mg_type_level, poly_time_sequence, coarsening_type, time_before_space,
                                                     space_time_level_first, operators, vanka);
    gmg->reinit();
  }

  // one global coarsening step of a structured block: every second vertex plane
  static Mesh coarsen(const Mesh &fine)
  {
    Mesh c = fine;
    for (int d = 0; d < 3; ++d) {
      if (fine.ncell[d] % 2) throw std::invalid_argument("STMGHierarchy: odd cell count, no coarser mesh");
      c.ncell[d] = fine.ncell[d] / 2;
    }
    if (!fine.vertices.empty()) {
      c.vertices.clear();
      const size_t nvx = fine.ncell[0] + 1, nvy = fine.ncell[1] + 1;
      for (int k = 0; k <= fine.ncell[2]; k += 2)
        for (int j = 0; j <= fine.ncell[1]; j += 2)
          for (int i = 0; i <= fine.ncell[0]; i += 2)
            for (int e = 0; e < 3; ++e) c.vertices.push_back(fine.vertices[3 * (i + nvx * (j + nvy * size_t(k))) + e]);
    }
    return c;
  }
};

} // namespace stfem
