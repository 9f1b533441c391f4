// The solver side of the Stokes slab problem (tests/tp_03stokes.cc:536-560, 840-1090) over the C-ABI: the two-variable block vector
// with the vector arithmetic of the Krylov solver, the relaxation smoother around the two-variable Vanka smoother, and the first-order
// time integrator of include/time_integrators.h:30-336 with a velocity force, for FE_Q(2)^3 x FE_Q(1) (BASELINE configs[4]).
// Control flow only: every vector operation, integral and operator application is a call into libstfem_hip.so.
#pragma once
#include "stmg.h"
#include "stokes.h"
#include "time_integrators.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace stfem {

// The scalar spaces behind the two variables: a velocity component is a FE_Q(2) function (with the velocity's strong constraints),
// the pressure a FE_Q(1) function without constraints.  Their contexts serve the vector arithmetic, the load vectors, the
// interpolation and the error norms of the Stokes vectors (VectorTools::* per variable in the reference).
struct StokesSpaces {
  std::shared_ptr<Context> q2, q1; // q1: the pressure space's context (FE_Q(1): a real one; FE_DGP(1): a carrier for the vector arithmetic)
  stfem_stokes_ctx *stokes;
  StokesSpaces(const Mesh &mesh, stfem_stokes_ctx *stokes_) : stokes(stokes_)
  {
    MatrixFreeOperator<3, 1, double> a(mesh, 2, 0.0, 1.0);
    q2 = a.context();
    stfem_ctx *pc = nullptr;
    check(stfem_stokes_pressure_ctx(stokes, &pc), "stfem_stokes_pressure_ctx");
    q1 = std::make_shared<Context>(pc, false);
    q1->degree = 1;
  }
  // z-slab partition (the mesh is this rank's slab, interface faces taken out of its Dirichlet mask): the operator leaves partial
  // sums in the interface planes of every velocity component and of the pressure (tests/test_gpu_stokes.py::
  // test_stokes_on_z_slabs_equals_whole_mesh); StokesSystem::vmult completes them with the scalar spaces' add-exchange
  // (stfem_halo_begin / end), inner products count owned planes only (stfem_dot_global) - as for the scalar operators
  void set_partition(const std::shared_ptr<Communicator> &comm, int lower_rank, int upper_rank)
  {
    for (auto &c : {q2, q1}) {
      c->comm = comm;
      c->lower_rank = lower_rank;
      c->upper_rank = upper_rank;
    }
  }
};

// BlockVectorT over a two-variable BlockSlice (LinearAlgebra::distributed::BlockVector with velocity and pressure blocks)
class StokesBlockVector {
public:
  StokesBlockVector() = default;
  void reinit(const std::shared_ptr<StokesSpaces> &spaces, stfem_stokes_ctx *stokes, const BlockSlice &slice)
  {
    spaces_ = spaces; stokes_ = stokes; slice_ = std::make_shared<BlockSlice>(slice);
    blocks_.clear();
    views_.clear();
    const size_t nu = size_t(stfem_stokes_n_velocity_dofs(stokes));
    for (unsigned i = 0; i < slice.n_blocks(); ++i) {
      const int var = int(slice.decompose(i)[1]);
      blocks_.emplace_back(stokes, var);
      BlockVectorT<double> v;
      if (var == 0) {
        void *ptrs[3] = {blocks_.back().data(), blocks_.back().data() + nu, blocks_.back().data() + 2 * nu};
        v.wrap(spaces->q2, ptrs, 3);
      } else {
        void *ptrs[1] = {blocks_.back().data()};
        v.wrap(spaces->q1, ptrs, 1);
      }
      views_.push_back(std::move(v));
    }
  }
  bool empty() const { return blocks_.empty(); }
  unsigned n_blocks() const { return unsigned(blocks_.size()); }
  std::vector<StokesVector> &blocks() { return blocks_; }
  const std::vector<StokesVector> &blocks() const { return blocks_; }
  // block i as a block vector of its scalar space (velocity: the three components)
  BlockVectorT<double> &view(unsigned i) { return views_.at(i); }
  const BlockVectorT<double> &view(unsigned i) const { return views_.at(i); }
  const std::shared_ptr<StokesSpaces> &spaces() const { return spaces_; }
  stfem_stokes_ctx *stokes() const { return stokes_; }
  const BlockSlice &slice() const { return *slice_; }

private:
  std::shared_ptr<StokesSpaces> spaces_;
  stfem_stokes_ctx *stokes_ = nullptr;
  std::shared_ptr<BlockSlice> slice_;
  std::vector<StokesVector> blocks_;
  std::vector<BlockVectorT<double>> views_;
};

inline void reinit_like(StokesBlockVector &v, const StokesBlockVector &x)
{
  if (v.empty()) v.reinit(x.spaces(), x.stokes(), x.slice());
}
// all blocks in one launch (stfem_axpby_many: the blocks of the two variables differ in length)
inline void axpby(double a, const StokesBlockVector &x, double b, StokesBlockVector &y, void *stream = nullptr)
{
  const unsigned nb = y.n_blocks();
  std::vector<int64_t> len(nb);
  std::vector<const void *> px(nb);
  std::vector<void *> py(nb);
  for (unsigned i = 0; i < nb; ++i) {
    if (x.blocks()[i].size() != y.blocks()[i].size()) throw Error(STFEM_ERR_SHAPE_MISMATCH, "axpby: block sizes");
    len[i] = int64_t(y.blocks()[i].size());
    px[i] = x.blocks()[i].data();
    py[i] = y.blocks()[i].data();
  }
  check(stfem_axpby_many(y.spaces()->q2->h, int(nb), len.data(), a, px.data(), b, py.data(), stream), "stfem_axpby_many");
}
inline void set_zero(StokesBlockVector &v, void *stream = nullptr) { axpby(0.0, v, 0.0, v, stream); }
inline double dot(const StokesBlockVector &a, const StokesBlockVector &b)
{
  double s = 0.0;
  for (unsigned i = 0; i < a.n_blocks(); ++i) s += dot(a.view(i), b.view(i));
  return s;
}
inline double norm(const StokesBlockVector &x) { return std::sqrt(dot(x, x)); }
// the Gram-Schmidt step of the Krylov solver: classical scheme, the k inner products of a pass in one launch per block
// (stfem_multi_dot on the block's scalar view, summed over the blocks on the host), the k updates in one launch per block
// (stfem_multi_axpy); a second pass when the first cancelled most of w, as for the scalar systems (time_integrators.h)
inline double orthogonalize(const std::vector<StokesBlockVector> &vs, unsigned k, StokesBlockVector &w, double *h)
{
  const unsigned nb = w.n_blocks();
  std::vector<const stfem_vec *> handles(k);
  std::vector<double> part(k), minus(k);
  auto pass = [&](double *hh) {
    for (unsigned i = 0; i < k; ++i) hh[i] = 0.0;
    for (unsigned b = 0; b < nb; ++b) {
      for (unsigned i = 0; i < k; ++i) handles[i] = vs[i].view(b).handle();
      check(stfem_multi_dot(w.view(b).context()->h, int(k), handles.data(), w.view(b).handle(), 0, part.data(), nullptr), "stfem_multi_dot");
      for (unsigned i = 0; i < k; ++i) hh[i] += part[i];
    }
    for (unsigned i = 0; i < k; ++i) minus[i] = -hh[i];
    for (unsigned b = 0; b < nb; ++b) {
      for (unsigned i = 0; i < k; ++i) handles[i] = vs[i].view(b).handle();
      check(stfem_multi_axpy(w.view(b).context()->h, int(k), minus.data(), handles.data(), w.view(b).handle(), nullptr), "stfem_multi_axpy");
    }
  };
  if (k == 0 || k > 240 || w.spaces()->q2->comm) { // (partitioned vectors: the modified scheme with reducing inner products)
    for (unsigned i = 0; i < k; ++i) {
      h[i] = dot(w, vs[i]);
      axpby(-h[i], vs[i], 1.0, w);
    }
    return norm(w);
  }
  const double before = dot(w, w);
  pass(h);
  double after = dot(w, w);
  if (!(after > 0.01 * before)) {
    std::vector<double> h2(k);
    pass(h2.data());
    for (unsigned i = 0; i < k; ++i) h[i] += h2[i];
    after = dot(w, w);
  }
  return std::sqrt(std::max(after, 0.0));
}

// SystemMatrixStokes on StokesBlockVector (the operator interface the solver consumes)
template <int dim, typename Number> class StokesSystem {
public:
  StokesSystem(const SystemMatrixStokes<dim, Number> &A, const std::shared_ptr<StokesSpaces> &spaces, stfem_stokes_ctx *stokes, const BlockSlice &slice)
    : A(A), spaces(spaces), stokes(stokes), slice(slice)
  {}
  void initialize_dof_vector(StokesBlockVector &v) const { v.reinit(spaces, stokes, slice); }
  void vmult(StokesBlockVector &dst, const StokesBlockVector &src, void *stream = nullptr) const
  {
    A.vmult(dst.blocks(), src.blocks(), stream);
    if (spaces->q2->comm) // partitioned: dst.compress(add) on every block through its scalar view
      for (unsigned b = 0; b < dst.n_blocks(); ++b) compress_add(*dst.view(b).context(), dst.view(b).handle(), stream);
  }

private:
  const SystemMatrixStokes<dim, Number> &A;
  std::shared_ptr<StokesSpaces> spaces;
  stfem_stokes_ctx *stokes;
  BlockSlice slice;
};

// PreconditionRelaxation around the two-variable Vanka smoother (stmg.h:1199-1238): n sweeps of x <- x + omega P^-1 (b - A x) from 0
template <typename System> class PreconditionRelaxationStokes {
public:
  PreconditionRelaxationStokes(const System &A, const PreconditionVankaStokes<double> &P, double omega, unsigned n_iterations)
    : A(A), P(P), omega(omega), n_iterations(n_iterations)
  {}
  void vmult(StokesBlockVector &dst, const StokesBlockVector &src, void *stream = nullptr) const
  {
    P.step(dst.blocks(), omega, false, src.blocks(), stream);
    for (unsigned it = 1; it < n_iterations; ++it) {
      reinit_like(res, src);
      A.vmult(res, dst, stream);
      axpby(1.0, src, -1.0, res, stream);
      P.step(dst.blocks(), omega, true, res.blocks(), stream);
    }
  }

private:
  const System &A;
  const PreconditionVankaStokes<double> &P;
  double omega;
  unsigned n_iterations;
  mutable StokesBlockVector res;
};

template <typename System>
double estimate_relaxation_stokes(const System &A, const PreconditionVankaStokes<double> &P, unsigned n_iterations = 20, double smoothing_range = 1.0);

// The geometric multigrid of the reference's Stokes runs (tests/tp_03stokes.cc:283-290, 484-770: coarsening sequence in space from
// MGTransferGlobalCoarseningTools::create_geometric_coarsening_sequence, one StokesMatrixFreeOperator / SystemMatrixStokes /
// PreconditionVanka per level, MGTwoLevelBlockTransfer with one space transfer per variable; include/stmg.h:1160-1419: GMG with
// Multigrid, MGSmootherPrecondition around PreconditionRelaxation(Vanka), MGCoarseGridApplySmoother, PreconditionMG) over the
// C-ABI: the velocity components are transferred as FE_Q(2) functions with the constraints of both levels, the pressure as a
// FE_Q(1) function (stfem_transfer_*), every level smooths with relaxation sweeps of the two-variable Vanka smoother.  The first
// constructor coarsens in space only (the temporal blocks are the same on every level); the second takes the reference's level
// sequence (tests/tp_03stokes.cc:294-326: MGType h, k, tau from get_mg_sequence) - a k level lowers the temporal degree, a tau
// level halves the time steps per slab, both with the time transfers of MGTwoLevelTransferTime applied per variable
// (include/stmg.h:557-600: MGTwoLevelBlockTransfer with the time matrices), and every level has its own temporal matrices
// (get_fe_time_weights_stokes of its degree, step count and step size) and block structure.
template <int dim> class GMGStokes {
public:
  struct AdditionalData {
    unsigned smoothing_steps = 1;      // PreconditionerGMGAdditionalData::smoothing_steps
    bool variable = true;              // MGSmootherPrecondition variable: 2^(max_level - level) steps on level `level`
    unsigned smoothing_degree = 1;     // sweeps of PreconditionRelaxation per step
    double relaxation = 0.0;           // its omega; 0: estimated per level (20 power iterations on P^-1 A), as the reference's default
  };
  // the finest mesh has mesh.ncell cells; level l < n_levels - 1 has them halved n_levels - 1 - l times
  GMGStokes(const Mesh &mesh, unsigned n_levels, double viscosity, const FullMatrix<double> &Alpha, const FullMatrix<double> &Beta,
            const BlockSlice &slice, const AdditionalData &data = AdditionalData(), const std::set<boundary_id> &weak_boundary_ids = {},
            bool dg_pressure = false)
    : data_(data)
  {
    if (n_levels < 1) throw std::invalid_argument("GMGStokes: at least one level");
    dg_ = dg_pressure;
    levels_.resize(n_levels);
    for (unsigned l = 0; l < n_levels; ++l) {
      levels_[l].halvings = n_levels - 1 - l;
      levels_[l].Alpha = Alpha;
      levels_[l].Beta = Beta;
      levels_[l].slice = slice;
      levels_[l].from_below = MGType::h;
    }
    build(mesh, viscosity, weak_boundary_ids, TimeStepType::CGP);
  }
  // mg_type_level[l]: how level l + 1 (finer) arises from level l, finest last, as get_mg_sequence returns it (h, k and tau; the
  // Stokes element has no p levels); poly_time_sequence: the temporal degrees, ascending (get_poly_mg_sequence); the finest level has
  // degree poly_time_sequence.back(), n_timesteps_at_once steps of size time_step_size per slab
  GMGStokes(const Mesh &mesh, const std::vector<MGType> &mg_type_level, const std::vector<unsigned> &poly_time_sequence, TimeStepType type,
            double time_step_size, unsigned n_timesteps_at_once, double viscosity, const AdditionalData &data = AdditionalData(),
            const std::set<boundary_id> &weak_boundary_ids = {}, bool dg_pressure = false)
    : data_(data)
  {
    dg_ = dg_pressure;
    const unsigned n_levels = unsigned(mg_type_level.size()) + 1;
    levels_.resize(n_levels);
    const auto blk = get_blk_indices(type, n_timesteps_at_once, 2, n_levels, mg_type_level, poly_time_sequence);
    auto p_mg = poly_time_sequence.rbegin();
    unsigned halvings = 0;
    for (unsigned l = n_levels; l-- > 0;) {
      Level &L = levels_[l];
      L.halvings = halvings;
      L.slice = blk[l];
      const auto w = get_fe_time_weights_stokes<double>(type, *p_mg, time_step_size, n_timesteps_at_once);
      L.Alpha = w[0];
      L.Beta = w[1];
      if (l == 0) break;
      L.from_below = mg_type_level[l - 1];
      switch (mg_type_level[l - 1]) {
        case MGType::h: ++halvings; break;
        case MGType::k: ++p_mg; break;
        case MGType::tau: n_timesteps_at_once /= 2; time_step_size *= 2; break;
        default: throw std::invalid_argument("GMGStokes: h, k and tau levels");
      }
    }
    build(mesh, viscosity, weak_boundary_ids, type);
  }
  const StokesSystem<dim, double> &finest_system() const { return *levels_.back().system; }
  const StokesMatrixFreeOperator<dim, double> &finest_operator() const { return *levels_.back().K; }
  const std::shared_ptr<StokesSpaces> &finest_spaces() const { return levels_.back().spaces; }
  double relaxation(unsigned level) const { return levels_.at(level).omega; }
  unsigned n_levels() const { return unsigned(levels_.size()); }
  // STFEM_MG_TIMING=1: wall time per level (smoothing, residual + transfers), with a device synchronisation around every section -
  // the sections of the reference's TimerOutput ("gmg" and below); the synchronisations cost the overlap of launches and kernels
  void print_timing(FILE *f) const
  {
    if (!timing_) return;
    for (unsigned l = 0; l < levels_.size(); ++l)
      std::fprintf(f, "level %u: smoothing %.3f s, residual and transfers %.3f s (%u cycles)\n", l, levels_[l].t_smooth, levels_[l].t_transfer, cycles_);
  }

  // PreconditionMG::vmult: copy_to_mg, one V-cycle from zero, copy_from_mg
  void vmult(StokesBlockVector &dst, const StokesBlockVector &src, void * = nullptr) const
  {
    TraceRange scope("gmg");
    const unsigned top = unsigned(levels_.size()) - 1;
    // (the caller's vectors may live on contexts of their own for the same mesh: view their blocks through this level's spaces)
    const unsigned nb = levels_[top].slice.n_blocks();
    for (unsigned b = 0; b < nb; ++b) axpby(1.0, foreign_view(src, b), 0.0, levels_[top].defect.view(b));
    level_v_step(top);
    ++cycles_;
    for (unsigned b = 0; b < nb; ++b) {
      BlockVectorT<double> d = foreign_view(dst, b);
      axpby(1.0, levels_[top].solution.view(b), 0.0, d);
    }
  }

private:
  struct Level {
    Mesh mesh;
    unsigned halvings = 0;            // of the finest mesh's cell counts
    FullMatrix<double> Alpha, Beta;   // the level's temporal matrices and block structure
    BlockSlice slice;
    MGType from_below = MGType::h;    // the transfer between this level and the one below
    FullMatrix<double> time_prolongation, time_restriction; // k / tau: one variable's blocks (BlockSlice(steps, 1, dofs) order)
    std::unique_ptr<StokesMatrixFreeOperator<dim, double>> K;
    std::shared_ptr<StokesSpaces> spaces;
    std::unique_ptr<SystemMatrixStokes<dim, double>> A;
    std::unique_ptr<StokesSystem<dim, double>> system;
    std::unique_ptr<PreconditionVankaStokes<double>> vanka;
    std::unique_ptr<PreconditionRelaxationStokes<StokesSystem<dim, double>>> relax;
    std::unique_ptr<MGTwoLevelTransfer<double>> tr_u, tr_p; // to the level below
    double omega = 1.0;
    mutable StokesBlockVector defect, solution, t, tmp;
    mutable double t_smooth = 0.0, t_transfer = 0.0;
  };
  struct Section { // (times one section of the cycle when STFEM_MG_TIMING is set)
    Section(bool on, double &acc) : on(on), acc(acc)
    {
      if (on) {
        (void)stfem_stream_synchronize(nullptr);
        t0 = std::chrono::steady_clock::now();
      }
    }
    ~Section()
    {
      if (on) {
        (void)stfem_stream_synchronize(nullptr);
        acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      }
    }
    bool on;
    double &acc;
    std::chrono::steady_clock::time_point t0;
  };
  void build(const Mesh &mesh, double viscosity, const std::set<boundary_id> &weak_boundary_ids, TimeStepType type)
  {
    const unsigned n_levels = unsigned(levels_.size());
    for (unsigned l = 0; l < n_levels; ++l) {
      Level &L = levels_[l];
      L.mesh = mesh;
      for (int d = 0; d < 3; ++d) {
        const int f = 1 << L.halvings;
        if (mesh.ncell[d] % f) throw std::invalid_argument("GMGStokes: the cell counts must be divisible by 2^(space levels - 1)");
        L.mesh.ncell[d] = mesh.ncell[d] / f;
      }
      L.K = std::make_unique<StokesMatrixFreeOperator<dim, double>>(L.mesh, 2, viscosity, weak_boundary_ids, std::set<boundary_id>(), 20.0, 10.0, 0.0, 0.0, 0.0,
                                                                   dg_);
      L.spaces = std::make_shared<StokesSpaces>(L.mesh, L.K->handle());
      L.A = std::make_unique<SystemMatrixStokes<dim, double>>(*L.K, L.Alpha, L.Beta, L.slice);
      L.system = std::make_unique<StokesSystem<dim, double>>(*L.A, L.spaces, L.K->handle(), L.slice);
      L.vanka = std::make_unique<PreconditionVankaStokes<double>>(*L.K, L.Alpha, L.Beta, L.slice);
      L.omega = data_.relaxation != 0.0 ? data_.relaxation : estimate_relaxation_stokes(*L.system, *L.vanka, 20, 1.0);
      L.relax = std::make_unique<PreconditionRelaxationStokes<StokesSystem<dim, double>>>(*L.system, *L.vanka, L.omega, data_.smoothing_degree);
      L.system->initialize_dof_vector(L.defect);
      L.system->initialize_dof_vector(L.solution);
      L.system->initialize_dof_vector(L.t);
      if (l == 0) continue;
      const Level &C = levels_[l - 1];
      if (L.from_below == MGType::h) {
        L.tr_u = std::make_unique<MGTwoLevelTransfer<double>>(L.spaces->q2, C.spaces->q2);
        if (!dg_) L.tr_p = std::make_unique<MGTwoLevelTransfer<double>>(L.spaces->q1, C.spaces->q1);
      } else { // (restrict = transposed prolongation: the reference's default, parameters.h:29)
        const MGTwoLevelTransferTime<double> tt(BlockSlice(L.slice.n_timesteps_at_once(), 1, L.slice.n_timedofs()),
                                                BlockSlice(C.slice.n_timesteps_at_once(), 1, C.slice.n_timedofs()), type, true, L.from_below);
        L.time_prolongation = tt.prolongation();
        L.time_restriction = tt.restriction();
      }
    }
  }
  // dst (+)= (matrix x identity) src on the blocks of every variable: matrix acts between the (step, time dof) blocks of one variable
  void time_transfer(const Level &D, StokesBlockVector &dst, const FullMatrix<double> &matrix, const Level &S, const StokesBlockVector &src) const
  {
    for (unsigned v = 0; v < 2; ++v) {
      const unsigned ncomp = v == 0 ? 3 : 1;
      const size_t len = v == 0 ? size_t(stfem_stokes_n_velocity_dofs(D.K->handle())) : 0;
      auto gather = [&](const Level &L, const StokesBlockVector &x) {
        std::vector<void *> ptrs;
        for (unsigned it = 0; it < L.slice.n_timesteps_at_once(); ++it)
          for (unsigned id = 0; id < L.slice.n_timedofs(); ++id)
            for (unsigned c = 0; c < ncomp; ++c) ptrs.push_back(x.blocks()[L.slice.index(it, v, id)].data() + c * len);
        return ptrs;
      };
      std::vector<void *> pd = gather(D, dst), ps = gather(S, src);
      BlockVectorT<double> vd, vs;
      const auto &ctx = v == 0 ? D.spaces->q2 : D.spaces->q1; // (the same mesh on both levels: one context serves both)
      vd.wrap(ctx, pd.data(), unsigned(pd.size()));
      vs.wrap(ctx, ps.data(), unsigned(ps.size()));
      std::vector<double> a(pd.size() * ps.size(), 0.0);
      for (unsigned i = 0; i < matrix.m(); ++i)
        for (unsigned j = 0; j < matrix.n(); ++j)
          for (unsigned c = 0; c < ncomp; ++c) a[size_t(i * ncomp + c) * ps.size() + j * ncomp + c] = matrix(i, j);
      check(stfem_tensorproduct_add(ctx->h, int(pd.size()), int(ps.size()), a.data(), vd.handle(), vs.handle(), nullptr), "GMGStokes: time transfer");
    }
  }
  BlockVectorT<double> foreign_view(const StokesBlockVector &x, unsigned b) const
  {
    const StokesSpaces &sp = *levels_.back().spaces;
    BlockVectorT<double> v;
    double *base = x.blocks()[b].data();
    if (levels_.back().slice.decompose(b)[1] == 0) {
      const size_t nu = size_t(stfem_stokes_n_velocity_dofs(levels_.back().K->handle()));
      void *ptrs[3] = {base, base + nu, base + 2 * nu};
      v.wrap(sp.q2, ptrs, 3);
    } else {
      void *ptrs[1] = {base};
      v.wrap(sp.q1, ptrs, 1);
    }
    return v;
  }
  // MGSmootherPrecondition::smooth / apply: u (= | +=) P (rhs - A u), `steps` times
  void smooth(unsigned level, bool from_zero) const
  {
    const Level &L = levels_[level];
    const unsigned steps = data_.smoothing_steps * (data_.variable ? 1u << (unsigned(levels_.size()) - 1 - level) : 1u);
    unsigned i = 0;
    if (from_zero) {
      L.relax->vmult(L.solution, L.defect);
      i = 1;
    }
    for (; i < steps; ++i) {
      reinit_like(L.tmp, L.defect);
      L.system->vmult(L.t, L.solution);
      axpby(1.0, L.defect, -1.0, L.t);
      L.relax->vmult(L.tmp, L.t);
      axpby(1.0, L.tmp, 1.0, L.solution);
    }
  }
  // Multigrid::level_v_step
  void level_v_step(unsigned level) const
  {
    const Level &L = levels_[level];
    if (level == 0) { // MGCoarseGridApplySmoother
      Section sec(timing_, L.t_smooth);
      smooth(0, true);
      return;
    }
    const Level &C = levels_[level - 1];
    {
      Section sec(timing_, L.t_smooth);
      smooth(level, true);
    }
    std::unique_ptr<Section> down(new Section(timing_, L.t_transfer));
    L.system->vmult(L.t, L.solution);
    axpby(1.0, L.defect, -1.0, L.t);
    set_zero(C.defect);
    const BlockSlice &slice = L.slice;
    const bool in_time = L.from_below != MGType::h; // MGTwoLevelBlockTransfer with the time matrices, or one space transfer per variable
    if (in_time) time_transfer(C, C.defect, L.time_restriction, L, L.t);
    for (unsigned b = 0; b < slice.n_blocks() && !in_time; ++b) { // restrict_and_add: block by block with its variable's transfer
      if (slice.decompose(b)[1] == 1 && dg_) {
        check(stfem_stokes_dgp_restrict(L.K->handle(), C.K->handle(), C.defect.blocks()[b].data(), L.t.blocks()[b].data(), 1, nullptr), "GMGStokes: restrict_and_add");
        continue;
      }
      const MGTwoLevelTransfer<double> &tr = slice.decompose(b)[1] == 0 ? *L.tr_u : *L.tr_p;
      check(stfem_transfer_restrict(tr.handle(), C.defect.view(b).handle(), L.t.view(b).handle(), 1, nullptr), "GMGStokes: restrict_and_add");
    }
    down.reset();
    level_v_step(level - 1);
    std::unique_ptr<Section> up(new Section(timing_, L.t_transfer));
    if (in_time) time_transfer(L, L.solution, L.time_prolongation, C, C.solution);
    for (unsigned b = 0; b < slice.n_blocks() && !in_time; ++b) {
      if (slice.decompose(b)[1] == 1 && dg_) {
        check(stfem_stokes_dgp_prolongate(L.K->handle(), C.K->handle(), L.solution.blocks()[b].data(), C.solution.blocks()[b].data(), 1, nullptr),
              "GMGStokes: prolongate_and_add");
        continue;
      }
      const MGTwoLevelTransfer<double> &tr = slice.decompose(b)[1] == 0 ? *L.tr_u : *L.tr_p;
      check(stfem_transfer_prolongate(tr.handle(), L.solution.view(b).handle(), C.solution.view(b).handle(), 1, nullptr), "GMGStokes: prolongate_and_add");
    }
    up.reset();
    Section sec(timing_, L.t_smooth);
    smooth(level, false);
  }
  bool timing_ = [] {
    const char *e = std::getenv("STFEM_MG_TIMING");
    return e && std::atoi(e) != 0;
  }();
  mutable unsigned cycles_ = 0;
  AdditionalData data_;
  bool dg_ = false;
  std::vector<Level> levels_;
};

// The relaxation parameter of PreconditionRelaxation when the reference leaves it at 0 (parameters.h:19, stmg.h:1207-1213): deal.II
// estimates the largest eigenvalue of P^-1 A with a power iteration and takes 2 / (alpha + beta), beta = 1.2 lambda, alpha =
// min(0.9 beta, 1) for smoothing_range <= 1 (see estimate_relaxation in stmg.h); the same for the two-variable system.
template <typename System>
double estimate_relaxation_stokes(const System &A, const PreconditionVankaStokes<double> &P, unsigned n_iterations, double smoothing_range)
{
  StokesBlockVector v, w, z;
  A.initialize_dof_vector(v);
  A.initialize_dof_vector(w);
  A.initialize_dof_vector(z);
  for (unsigned b = 0; b < v.n_blocks(); ++b) {
    const size_t n = v.blocks()[b].size();
    std::vector<double> guess(n);
    double mean = 0.0;
    for (size_t i = 0; i < n; ++i) mean += double(i % 11);
    mean /= double(n);
    for (size_t i = 0; i < n; ++i) guess[i] = double(i % 11) - mean;
    v.blocks()[b].copy_from_host(guess);
  }
  axpby(0.0, v, 1.0 / norm(v), v);
  double lambda = 0.0;
  for (unsigned it = 0; it < n_iterations; ++it) {
    A.vmult(z, v);
    P.vmult(w.blocks(), z.blocks());
    lambda = dot(v, w);
    const double nw = norm(w);
    if (!(nw > 0)) break;
    axpby(1.0 / nw, w, 0.0, v);
  }
  lambda = std::abs(lambda);
  if (!(lambda > 0) || !std::isfinite(lambda)) return 1.0;
  const double beta = 1.2 * lambda, alpha = smoothing_range > 1.0 ? beta / smoothing_range : std::min(0.9 * beta, 1.0);
  return 2.0 / (alpha + beta);
}

// ErrorCalculator (include/exact_solution.h:503-649) for the pressure variable in either pressure space: the temporal Lagrange
// combination of the pressure blocks (evaluate_numerical_solution), then stfem_stokes_pressure_difference with QGauss(nq)^3
class PressureErrorCalculator {
public:
  PressureErrorCalculator(TimeStepType type, unsigned time_degree, int nq_space, const std::shared_ptr<StokesSpaces> &spaces, const PointFunction &exact)
    : type(type), time_degree(time_degree), nq(nq_space), spaces(spaces), exact(exact), nodes(time_points(type, time_degree)), tq(time_degree + 1),
      tw(time_degree + 1)
  {
    check(stfem_gauss_rule(int(time_degree + 1), tq.data(), tw.data()), "stfem_gauss_rule");
    numeric.reinit(spaces->q1, 1);
    const size_t ncells = size_t(stfem_n_cells(spaces->q2->h));
    qpoints.resize(ncells * nq * nq * nq * 3);
    check(stfem_stokes_pressure_quadrature_points(spaces->stokes, nq, qpoints.data()), "stfem_stokes_pressure_quadrature_points");
  }
  // {L2^2 contribution, Linfty} of the slab; x: the pressure blocks of the time dofs (views on the pressure space), prev: the previous one
  std::array<double, 2> evaluate_error(double time, double time_step, const BlockVectorT<double> &x, const BlockVectorT<double> &prev)
  {
    std::array<double, 2> err{0.0, -1.0};
    const unsigned nt_dofs = type == TimeStepType::DG ? time_degree + 1 : time_degree;
    std::vector<double> pe;
    for (unsigned q = 0; q < tq.size(); ++q) {
      const std::vector<double> L = lagrange_values(nodes, tq[q]);
      set_zero(numeric);
      if (type == TimeStepType::DG) {
        for (unsigned i = 0; i < nt_dofs; ++i) axpby(L[i], block_view(x, i), 1.0, numeric);
      } else {
        axpby(L[0], prev, 1.0, numeric);
        for (unsigned i = 1; i <= time_degree; ++i) axpby(L[i], block_view(x, i - 1), 1.0, numeric);
      }
      exact(time + tq[q] * time_step, qpoints, pe);
      double out[2];
      check(stfem_stokes_pressure_difference(spaces->stokes, nq, static_cast<const double *>(stfem_vector_block(numeric.handle(), 0)), pe.data(), out, nullptr),
            "stfem_stokes_pressure_difference");
      err[0] += time_step * tw[q] * out[0];
      err[1] = std::max(err[1], out[1]);
    }
    return err;
  }

private:
  TimeStepType type;
  unsigned time_degree;
  int nq;
  std::shared_ptr<StokesSpaces> spaces;
  PointFunction exact;
  std::vector<double> nodes, tq, tw, qpoints;
  BlockVectorT<double> numeric;
};

// A vector function of (x, t) at a list of points: out[c][i] = f_c(points[3 i .. 3 i + 2], t)
using VectorPointFunction = std::function<void(double time, const std::vector<double> &points, std::array<std::vector<double>, 3> &out)>;

// include/time_integrators.h:30-336 for the two-variable system: rhs = rhs_matrix (prev_u, prev_p) + time quadrature of the velocity
// force (assemble_force per variable, 73-111: the pressure load is zero), FGMRES on the slab system, the pressure of every time
// dof shifted to zero mean afterwards (tests/tp_03stokes.cc:1047-1062).  Alpha_1 / Gamma_1: the ONE-step scalar temporal matrices.
template <int dim, typename System, typename Preconditioner> class TimeIntegratorStokes {
public:
  TimeIntegratorStokes(TimeStepType type, unsigned time_degree, const FullMatrix<double> &Alpha_1, const FullMatrix<double> &Gamma_1,
                       double gmres_tolerance, const System &matrix, const Preconditioner &preconditioner,
                       const SystemMatrixStokes<dim, double> &rhs_matrix, const VectorPointFunction &force, bool zero_mean_pressure,
                       double abstol = 1e-12, unsigned max_steps = 400)
    : type(type), time_degree(time_degree), quad_time(time_points(type, time_degree)), Alpha(Alpha_1), Gamma(Gamma_1),
      solver(max_steps, abstol, gmres_tolerance, 200), preconditioner(preconditioner), matrix(matrix), rhs_matrix(rhs_matrix), force(force),
      nt_dofs(type == TimeStepType::DG ? time_degree + 1 : time_degree), zero_mean(zero_mean_pressure)
  {
    if (const char *e = std::getenv("STFEM_FGMRES_VERBOSE")) solver.verbose = unsigned(std::atoi(e));
  }

  // x, rhs: the slab's blocks; prev: one (velocity, pressure) pair (BlockSlice(1, 2, 1))
  void solve(StokesBlockVector &x, const StokesBlockVector &prev, StokesBlockVector &rhs, double time, double time_step)
  {
    TraceRange scope("step");
    const StokesSpaces &sp = *x.spaces();
    const BlockSlice &slice = x.slice();
    set_zero(rhs);
    rhs_matrix.vmult_slice_add(rhs.blocks(), prev.blocks());
    // assemble_force: Alpha is diagonal (time quadrature = support points)
    if (qpoints.empty()) {
      qpoints.resize(size_t(stfem_n_cells(sp.q2->h)) * 27 * 3);
      check(stfem_quadrature_points(sp.q2->h, 3, qpoints.data()), "stfem_quadrature_points");
      load.reinit(sp.q2, 3);
    }
    std::array<std::vector<double>, 3> fq;
    for (unsigned j = 0; j < quad_time.size(); ++j) {
      force(time + time_step * quad_time[j], qpoints, fq);
      for (int c = 0; c < 3; ++c) check(stfem_integrate_rhs(sp.q2->h, 3, fq[c].data(), load.handle(), c, nullptr), "stfem_integrate_rhs");
      auto add = [&](unsigned timedof, double w) { axpby(w, load, 1.0, rhs.view(slice.index(0, 0, timedof))); };
      if (type == TimeStepType::DG) add(j, Alpha(j, j));
      else if (j == 0)
        for (unsigned i = 0; i < nt_dofs; ++i) add(i, -Gamma(i, 0));
      else add(j - 1, Alpha(j - 1, j - 1));
    }
    for (unsigned i = 0; i < x.n_blocks(); ++i) // extrapolate (time_integrators.h:184-194): every time dof starts from the previous solution
      axpby(1.0, prev.view(slice.decompose(i)[1]), 0.0, x.view(i));
    (void)dot(rhs.view(0), rhs.view(0)); // (synchronises: the clock below sees the Krylov solve alone)
    const auto t0 = std::chrono::steady_clock::now();
    solver.solve(matrix, x, rhs, preconditioner);
    (void)dot(x.view(0), x.view(0));
    solver_seconds_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (zero_mean) {
      if (!weights.handle()) { // mean(p) = weights . p / |Omega|, p -= mean * ones (VectorTools::compute_mean_value / add_constant)
        weights.reinit(sp.q1, 1);
        ones.reinit(sp.q1, 1);
        std::vector<std::vector<double>> h1(1, std::vector<double>(ones.block_size())), hw(1, std::vector<double>(ones.block_size()));
        check(stfem_stokes_pressure_mean_vectors(sp.stokes, h1[0].data(), hw[0].data(), &volume), "stfem_stokes_pressure_mean_vectors");
        ones.copy_from_host(h1);
        weights.copy_from_host(hw);
      }
      for (unsigned a = 0; a < nt_dofs; ++a) {
        BlockVectorT<double> &p = x.view(slice.index(0, 1, a));
        axpby(-dot(weights, p) / volume, ones, 1.0, p);
      }
    }
  }
  unsigned last_step() const { return solver.last_step(); }
  double solver_seconds() const { return solver_seconds_; } // wall time of the FGMRES solves so far (without the right-hand sides)

private:
  double solver_seconds_ = 0.0;
  TimeStepType type;
  unsigned time_degree;
  std::vector<double> quad_time;
  const FullMatrix<double> &Alpha, &Gamma;
  SolverFGMRES<double, StokesBlockVector> solver;
  const Preconditioner &preconditioner;
  const System &matrix;
  const SystemMatrixStokes<dim, double> &rhs_matrix;
  VectorPointFunction force;
  unsigned nt_dofs;
  bool zero_mean;
  std::vector<double> qpoints;
  BlockVectorT<double> load, weights, ones;
  double volume = 1.0;
};

} // namespace stfem
