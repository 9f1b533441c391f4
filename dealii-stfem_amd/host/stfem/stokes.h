// Host-side mirror of the reference's Stokes operators (include/operators.h:666-868
// SystemMatrixStokes, 1193-1766 StokesMatrixFreeOperator, 1768-1951 StokesNitscheMatrixFreeOperator;
// include/fe_time.h:901-1221 BlockSlice, 1242-1285 get_fe_time_weights_stokes) on the C-ABI (stfem_stokes_*):
// cell loop and, for weak boundary ids, the boundary-face loop of the linear operator.
#pragma once
#include "fe_time.h"
#include "operators.h"

#include <array>
#include <functional>
#include <limits>
#include <map>
#include <set>

namespace stfem {

// fe_time.h:901-1010 block_indexing / BlockSlice: block <-> (timestep, variable, timedof)
class BlockSlice {
public:
  BlockSlice(unsigned n_timesteps_at_once = 1, unsigned n_variables = 1, unsigned n_timedofs = 1, bool variable_major = true)
    : nts_(n_timesteps_at_once), nv_(n_variables), ntd_(n_timedofs), variable_major_(variable_major)
  {}
  unsigned index(unsigned timestep, unsigned variable, unsigned timedof) const
  {
    return variable_major_ ? timestep * (nv_ * ntd_) + variable * ntd_ + timedof
                           : timestep * (nv_ * ntd_) + timedof * nv_ + variable;
  }
  std::array<unsigned, 3> decompose(unsigned i) const
  {
    const unsigned ts = i / (nv_ * ntd_), r = i % (nv_ * ntd_);
    return variable_major_ ? std::array<unsigned, 3>{{ts, r / ntd_, r % ntd_}}
                           : std::array<unsigned, 3>{{ts, r % nv_, r / nv_}};
  }
  unsigned n_timesteps_at_once() const { return nts_; }
  unsigned n_variables() const { return nv_; }
  unsigned n_timedofs() const { return ntd_; }
  unsigned n_blocks() const { return nts_ * nv_ * ntd_; }
  bool variable_major() const { return variable_major_; }

private:
  unsigned nts_, nv_, ntd_;
  bool variable_major_;
};

// fe_time.h:1242-1285: {Alpha, Beta, Gamma, Zeta} in the (variable, time dof) block structure: the
// scalar matrices of get_fe_time_weights scattered over the two variables; no pressure-pressure
// block in Alpha, only velocity-velocity in Beta; Gamma / Zeta (right-hand side) on the velocity
// rows, Gamma for cG on the pressure rows too (1275-1282)
template <typename Number>
std::array<FullMatrix<Number>, 4> get_fe_time_weights_stokes(TimeStepType type, unsigned r, double time_step_size,
                                                             unsigned n_timesteps_at_once = 1)
{
  const auto tw = get_fe_time_weights<Number>(type, r, time_step_size, n_timesteps_at_once);
  const unsigned n = tw[0].m(), nt = n / n_timesteps_at_once;
  BlockSlice s(n_timesteps_at_once, 2, nt);
  std::array<FullMatrix<Number>, 4> ret{{FullMatrix<Number>(2 * n, 2 * n), FullMatrix<Number>(2 * n, 2 * n),
                                         FullMatrix<Number>(2 * n, tw[2].n()), FullMatrix<Number>(2 * n, tw[3].n())}};
  auto idx = [&](unsigned v, unsigned k) { return s.index(k / nt, v, k % nt); };
  for (unsigned a = 0; a < n; ++a) {
    for (unsigned b = 0; b < n; ++b) {
      for (unsigned iv = 0; iv < 2; ++iv)
        for (unsigned jv = 0; jv < 2; ++jv)
          if (!(iv == 1 && jv == 1)) ret[0](idx(iv, a), idx(jv, b)) = tw[0](a, b);
      ret[1](idx(0, a), idx(0, b)) = tw[1](a, b);
    }
    for (unsigned q = 0; q < tw[2].n(); ++q) {
      ret[2](idx(0, a), q) = tw[2](a, q);
      if (type == TimeStepType::CGP) ret[2](idx(1, a), q) = tw[2](a, q);
    }
    for (unsigned q = 0; q < tw[3].n(); ++q) ret[3](idx(0, a), q) = tw[3](a, q);
  }
  return ret;
}

// one device vector of a Stokes operator (velocity: 3 * n_velocity doubles, pressure: n_pressure)
class StokesVector {
public:
  StokesVector() = default;
  StokesVector(stfem_stokes_ctx *c, int variable) : c_(c), variable_(variable)
  {
    check(stfem_stokes_vector_create(c, variable, &d_), "stfem_stokes_vector_create");
    n_ = variable == 0 ? 3 * size_t(stfem_stokes_n_velocity_dofs(c)) : size_t(stfem_stokes_n_pressure_dofs(c));
  }
  StokesVector(StokesVector &&o) noexcept : c_(o.c_), variable_(o.variable_), d_(o.d_), n_(o.n_) { o.d_ = nullptr; }
  StokesVector &operator=(StokesVector &&o) noexcept
  {
    std::swap(c_, o.c_); std::swap(variable_, o.variable_); std::swap(d_, o.d_); std::swap(n_, o.n_);
    return *this;
  }
  StokesVector(const StokesVector &) = delete;
  ~StokesVector() { if (d_) stfem_stokes_vector_destroy(c_, d_); }
  double *data() const { return d_; }
  size_t size() const { return n_; }
  void copy_from_host(const std::vector<double> &h)
  {
    if (h.size() != n_) throw std::invalid_argument("StokesVector size mismatch");
    check(stfem_stokes_vector_upload(c_, variable_, d_, h.data()), "stfem_stokes_vector_upload");
  }
  std::vector<double> copy_to_host() const
  {
    std::vector<double> h(n_);
    check(stfem_stokes_vector_download(c_, variable_, d_, h.data()), "stfem_stokes_vector_download");
    return h;
  }

private:
  stfem_stokes_ctx *c_ = nullptr;
  int variable_ = 0;
  double *d_ = nullptr;
  size_t n_ = 0;
};

// boundary ids of the structured block: 2 d + s (direction d, side s), as deal.II colorizes a hyper_rectangle
using boundary_id = unsigned;

// operators.h:1193-1766, linear operator: cell loop + boundary-face loop for the weak (Nitsche) ids.  Same constructor
// arguments after the mesh as the reference (1199-1212); delta0 != 0 (CIP interior faces) and the nonlinear treatments throw.
template <int dim, typename Number> class StokesMatrixFreeOperator {
  static_assert(dim == 3 && std::is_same<Number, double>::value, "3D, fp64");

public:
  using BlockVectorType = std::vector<StokesVector>; // {velocity, pressure}

  StokesMatrixFreeOperator(const Mesh &mesh, unsigned velocity_degree, Number viscosity, const std::set<boundary_id> &weak_boundary_ids = {},
                           const std::set<boundary_id> &outflow_boundary_ids = {}, Number penalty1 = 20, Number penalty2 = 10,
                           Number /*outflow_penalty*/ = 0.0, Number delta0 = 0.0, Number /*delta1*/ = 0.0, bool dg_pressure = false)
  {
    // dg_pressure: FE_DGP(degree - 1) instead of FE_Q(degree - 1) for the pressure (the reference chooses the element of its
    // second DoFHandler, tests/tp_03stokes.cc:83-86: dGPressure)
    if (delta0 != 0.0) throw Error(STFEM_ERR_UNSUPPORTED, "StokesMatrixFreeOperator: the CIP face term (delta0 != 0) is not built");
    create(mesh, velocity_degree, viscosity, dg_pressure);
    int weak = 0, outflow = 0;
    for (boundary_id f : weak_boundary_ids) weak |= 1 << f;
    for (boundary_id f : outflow_boundary_ids) outflow |= 1 << f;
    if (weak || outflow) check(stfem_stokes_set_weak_boundaries(h_, weak, outflow, penalty1, penalty2), "stfem_stokes_set_weak_boundaries");
  }

private:
  void create(const Mesh &mesh, unsigned velocity_degree, Number viscosity, bool dg_pressure)
  {
    stfem_mesh_desc md{};
    for (int d = 0; d < 3; ++d) {
      md.ncell[d] = mesh.ncell[d];
      md.lower[d] = mesh.lower[d];
      md.upper[d] = mesh.upper[d];
    }
    md.vertices = mesh.vertices.empty() ? nullptr : mesh.vertices.data();
    md.dirichlet_mask = mesh.dirichlet_mask;
    md.device = mesh.device;
    check(stfem_stokes_create_ex(&md, int(velocity_degree), dg_pressure ? 1 : 0, viscosity, &h_), "stfem_stokes_create");
  }

public:
  ~StokesMatrixFreeOperator() { stfem_stokes_destroy(h_); }
  StokesMatrixFreeOperator(const StokesMatrixFreeOperator &) = delete;

  void initialize_dof_vector(BlockVectorType &vec) const // operators.h:1254-1262
  {
    vec.clear();
    vec.emplace_back(h_, 0);
    vec.emplace_back(h_, 1);
  }
  void initialize_dof_vector(StokesVector &vec, unsigned variable) const { vec = StokesVector(h_, int(variable)); }

  void vmult(BlockVectorType &dst, const BlockVectorType &src, void *stream = nullptr) const
  {
    check(stfem_stokes_vmult(h_, dst.at(0).data(), dst.at(1).data(), src.at(0).data(), src.at(1).data(), stream),
          "StokesMatrixFreeOperator::vmult");
  }
  // the MassMatrixType of SystemMatrixStokes (vector mass)
  void mass_vmult(StokesVector &dst, const StokesVector &src, void *stream = nullptr) const
  {
    check(stfem_stokes_mass_vmult(h_, dst.data(), src.data(), stream), "vector mass vmult");
  }
  unsigned long long m() const { return 3ull * stfem_stokes_n_velocity_dofs(h_) + stfem_stokes_n_pressure_dofs(h_); }
  stfem_stokes_ctx *handle() const { return h_; }

private:
  stfem_stokes_ctx *h_ = nullptr;
};

// operators.h:1768-1951: the right-hand-side functional of the weakly imposed Dirichlet data.  It shares the geometry
// of the StokesMatrixFreeOperator it is built from (the reference builds a second MatrixFree from the same arguments).
template <int dim, typename Number> class StokesNitscheMatrixFreeOperator {
public:
  using BlockVectorType = std::vector<StokesVector>;
  using Function = std::function<std::array<Number, 3>(const std::array<Number, 3> &)>; // Function<dim, Number>::vector_value

  explicit StokesNitscheMatrixFreeOperator(const StokesMatrixFreeOperator<dim, Number> &op) : op_(op)
  {
    points_.resize(3 * size_t(stfem_stokes_n_face_points(op.handle())));
    if (!points_.empty()) check(stfem_stokes_face_points(op.handle(), points_.data()), "stfem_stokes_face_points");
  }
  // operators.h:1801-1807.  One function for all weak faces here (the reference maps boundary ids to functions; the
  // points of the faces come in ascending face order, so a caller with several functions can switch on the point)
  void set_dirichlet_functions(const Function &g) const { g_ = g; }
  void initialize_dof_vector(BlockVectorType &vec) const { op_.initialize_dof_vector(vec); }
  // operators.h:1833-1849: dst += the boundary integrals of g; nothing without Dirichlet functions
  void vmult(BlockVectorType &dst, void *stream = nullptr) const
  {
    if (!g_ || points_.empty()) return;
    std::vector<Number> gq(points_.size());
    for (size_t q = 0; q < points_.size() / 3; ++q) {
      const auto v = g_({{points_[3 * q], points_[3 * q + 1], points_[3 * q + 2]}});
      for (int e = 0; e < 3; ++e) gq[3 * q + e] = v[e];
    }
    check(stfem_stokes_nitsche_rhs(op_.handle(), gq.data(), dst.at(0).data(), dst.at(1).data(), stream), "StokesNitscheMatrixFreeOperator::vmult");
  }
  unsigned long long m() const { return op_.m(); }

private:
  const StokesMatrixFreeOperator<dim, Number> &op_;
  std::vector<double> points_;
  mutable Function g_;
};

// operators.h:666-868; blocks in BlockSlice order
template <int dim, typename Number> class SystemMatrixStokes {
public:
  using BlockVectorType = std::vector<StokesVector>;

  SystemMatrixStokes(const StokesMatrixFreeOperator<dim, Number> &K, const FullMatrix<Number> &Alpha_,
                     const FullMatrix<Number> &Beta_, const BlockSlice &blk_slice_)
    : K(K), Alpha(Alpha_), Beta(Beta_), blk_slice(blk_slice_)
  {
    if (Alpha.m() != blk_slice.n_blocks() || (Alpha.n() != Alpha.m() && Alpha.n() != 1) || Beta.m() != Alpha.m() ||
        Beta.n() != Alpha.n())
      throw std::invalid_argument("Alpha/Beta do not match the block slice");
  }
  void initialize_dof_vector(BlockVectorType &vec) const // operators.h:802-812
  {
    vec.clear();
    for (unsigned i = 0; i < blk_slice.n_blocks(); ++i) vec.emplace_back(K.handle(), int(blk_slice.decompose(i)[1]));
  }
  void vmult(BlockVectorType &dst, const BlockVectorType &src, void *stream = nullptr) const
  {
    const unsigned nb = blk_slice.n_blocks();
    if (dst.size() != nb || src.size() != nb) throw Error(STFEM_ERR_SHAPE_MISMATCH, "SystemMatrixStokes::vmult");
    std::vector<double *> d(nb);
    std::vector<const double *> s(nb);
    for (unsigned i = 0; i < nb; ++i) { d[i] = dst[i].data(); s[i] = src[i].data(); }
    check(stfem_stokes_st_vmult(K.handle(), int(blk_slice.n_timesteps_at_once()), int(blk_slice.n_timedofs()),
                                blk_slice.variable_major() ? 1 : 0, Alpha.data(), Beta.data(), d.data(), s.data(), stream),
          "SystemMatrixStokes::vmult");
  }
  // operators.h:708-745, as the reference has it: its scatter overload (operators.h:111-123) reads j = index(it, v, id),
  // i = index(jt, v, jd) - the result of source time dof (it, id) only reaches the destination blocks of the SAME time dof,
  // weighted with the entries of row j summed over (jt, jd).  Not a transpose; reproduced as one vmult with those matrices.
  void Tvmult(BlockVectorType &dst, const BlockVectorType &src, void *stream = nullptr) const
  {
    const unsigned nb = blk_slice.n_blocks(), ns = blk_slice.n_timesteps_at_once(), nt = blk_slice.n_timedofs();
    if (dst.size() != nb || src.size() != nb || Alpha.n() != nb) throw Error(STFEM_ERR_SHAPE_MISMATCH, "SystemMatrixStokes::Tvmult");
    const double eps10 = 10 * std::numeric_limits<Number>::epsilon();
    FullMatrix<Number> Ae(nb, nb), Be(nb, nb);
    for (unsigned it = 0; it < ns; ++it)
      for (unsigned id = 0; id < nt; ++id) {
        const unsigned col = blk_slice.index(it, 0, id);
        for (unsigned v = 0; v < 2; ++v) {
          const unsigned j = blk_slice.index(it, v, id);
          for (unsigned jt = 0; jt < ns; ++jt)
            for (unsigned jd = 0; jd < nt; ++jd) {
              const unsigned i = blk_slice.index(jt, v, jd);
              if (std::abs(Alpha(j, i)) > eps10) Ae(j, col) += Alpha(j, i);
              if (v == 0 && std::abs(Beta(j, i)) > eps10) Be(j, col) += Beta(j, i);
            }
        }
      }
    std::vector<double *> d(nb);
    std::vector<const double *> s(nb);
    for (unsigned i = 0; i < nb; ++i) { d[i] = dst[i].data(); s[i] = src[i].data(); }
    check(stfem_stokes_st_vmult(K.handle(), int(ns), int(nt), blk_slice.variable_major() ? 1 : 0, Ae.data(), Be.data(), d.data(), s.data(), stream),
          "SystemMatrixStokes::Tvmult");
  }
  // n x 1 case for the right-hand side (operators.h:748-781): Alpha, Beta are n x 1 here and src is one
  // (velocity, pressure) pair; dst is accumulated into
  void vmult_slice_add(BlockVectorType &dst, const BlockVectorType &src, void *stream = nullptr) const
  {
    const unsigned nb = blk_slice.n_blocks();
    if (dst.size() != nb || src.size() != 2 || Alpha.n() != 1) throw Error(STFEM_ERR_SHAPE_MISMATCH, "vmult_slice_add");
    std::vector<double *> d(nb);
    for (unsigned i = 0; i < nb; ++i) d[i] = dst[i].data();
    check(stfem_stokes_st_vmult_slice_add(K.handle(), int(blk_slice.n_timesteps_at_once()), int(blk_slice.n_timedofs()),
                                          blk_slice.variable_major() ? 1 : 0, Alpha.data(), Beta.data(), d.data(),
                                          src[0].data(), src[1].data(), stream),
          "SystemMatrixStokes::vmult_slice_add");
  }
  unsigned long long m() const { return (unsigned long long)(blk_slice.n_blocks() / 2) * K.m(); }

private:
  const StokesMatrixFreeOperator<dim, Number> &K;
  const FullMatrix<Number> &Alpha;
  const FullMatrix<Number> &Beta;
  BlockSlice blk_slice;
};

// PreconditionVanka in its block form (include/stmg.h:626-738, 832-872) as tests/tp_03stokes.cc:537-540, 714-726 creates it for the
// Stokes levels: the assembled Stokes and mass matrices restricted to every cell's velocity and pressure DoFs, combined with
// Alpha / Beta over the blocks of the BlockSlice (K_mask empty, M_mask(0, 0) only), inverted.  The assembled matrices and DoF
// handlers of the reference's constructor are not needed: the blocks follow from the operator (stfem_stokes_vanka_create).
template <typename Number> class PreconditionVankaStokes {
  static_assert(std::is_same<Number, double>::value, "fp64");

public:
  using BlockVectorType = std::vector<StokesVector>;
  template <int dim>
  PreconditionVankaStokes(const StokesMatrixFreeOperator<dim, Number> &K, const FullMatrix<Number> &Alpha, const FullMatrix<Number> &Beta,
                          const BlockSlice &blk_slice)
    : nb_(blk_slice.n_blocks())
  {
    if (Alpha.m() != nb_ || Alpha.n() != nb_ || Beta.m() != nb_ || Beta.n() != nb_) throw std::invalid_argument("Alpha/Beta do not match the block slice");
    std::vector<int32_t> var(nb_);
    for (unsigned i = 0; i < nb_; ++i) var[i] = int32_t(blk_slice.decompose(i)[1]);
    stfem_stokes_vanka *v = nullptr;
    const int rc = stfem_stokes_vanka_create(K.handle(), int(nb_), var.data(), Alpha.data(), Beta.data(), &v);
    if (rc != STFEM_OK) throw Error(rc, std::string("stfem_stokes_vanka_create: ") + stfem_stokes_vanka_last_error());
    v_.reset(v, stfem_stokes_vanka_destroy);
  }
  void vmult(BlockVectorType &dst, const BlockVectorType &src, void *stream = nullptr) const { step(dst, 1.0, false, src, stream); }
  void smooth(BlockVectorType &u, const BlockVectorType &rhs) const { vmult(u, rhs); }
  // dst = (accumulate ? dst : 0) + omega * vmult(src): the step of the PreconditionRelaxation around the smoother (stmg.h:1199-1238)
  void step(BlockVectorType &dst, double omega, bool accumulate, const BlockVectorType &src, void *stream = nullptr) const
  {
    if (dst.size() != nb_ || src.size() != nb_) throw Error(STFEM_ERR_SHAPE_MISMATCH, "PreconditionVankaStokes::vmult");
    std::vector<double *> d(nb_);
    std::vector<const double *> s(nb_);
    for (unsigned i = 0; i < nb_; ++i) { d[i] = dst[i].data(); s[i] = src[i].data(); }
    const int rc = stfem_stokes_vanka_step(v_.get(), d.data(), omega, accumulate ? 1 : 0, s.data(), stream);
    if (rc != STFEM_OK) throw Error(rc, std::string("PreconditionVankaStokes::vmult: ") + stfem_stokes_vanka_last_error());
  }
  int n_classes() const { return stfem_stokes_vanka_n_classes(v_.get()); }

private:
  unsigned nb_;
  std::shared_ptr<stfem_stokes_vanka> v_;
};

} // namespace stfem
