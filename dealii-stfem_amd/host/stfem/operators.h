// Host-side mirror of the reference's operator interface (include/operators.h) on the C-ABI:
// same class names, method names, argument meaning and error behaviour (exceptions instead of
// deal.II Assert/AssertThrow), so a caller written against the reference's duck-typed operator
// concept (vmult / Tvmult / vmult_slice(_add) / initialize_dof_vector / m / n) compiles against
// these classes.  Everything heavy happens behind libstfem_hip.so.
#pragma once
#include "fe_time.h"
#include "types.h"

#include <cmath>
#include <type_traits>

namespace stfem {

// src.update_ghost_values() / dst.compress(add) of a partitioned context (no-ops otherwise)
inline void ghost_update(const Context &c, stfem_vec *v, void *stream)
{
  if (c.partitioned())
    Communicator::check_comm(stfem_ghost_update(c.h, c.comm->handle(), v, c.lower_rank, c.upper_rank, stream),
                             "stfem_ghost_update");
}
inline void compress_add(const Context &c, stfem_vec *v, void *stream)
{
  if (!c.partitioned()) return;
  Communicator::check_comm(stfem_halo_begin(c.h, c.comm->handle(), v, c.lower_rank, c.upper_rank, stream), "stfem_halo_begin");
  Communicator::check_comm(stfem_halo_end(c.h, c.comm->handle(), v, stream), "stfem_halo_end");
}
// BlockVector::operator* (an MPI_Allreduce in the reference): owned entries only, summed over the ranks
template <typename Number> double dot(const BlockVectorT<Number> &a, const BlockVectorT<Number> &b, void *stream = nullptr)
{
  const Context &c = *a.context();
  double out = 0.0;
  if (c.comm)
    Communicator::check_comm(stfem_dot_global(c.h, c.comm->handle(), a.handle(), b.handle(), c.n_owned(), &out, stream),
                             "stfem_dot_global");
  else
    check(stfem_dot(c.h, a.handle(), b.handle(), 0, &out, stream), "stfem_dot");
  return out;
}

// Structured hexahedral mesh of one rank (GridGenerator::subdivided_hyper_rectangle +
// refine_global [+ distort_random], tests/tp_01.cc:82-90) and zero Dirichlet boundary ids.
struct Mesh {
  int ncell[3] = {1, 1, 1};
  double lower[3] = {0, 0, 0}, upper[3] = {1, 1, 1};
  std::vector<double> vertices; // empty = Cartesian box
  int dirichlet_mask = 63;
  int device = 0;

  void distort_random(double factor, uint64_t seed = 5489)
  {
    vertices.resize(size_t(ncell[0] + 1) * (ncell[1] + 1) * (ncell[2] + 1) * 3);
    check(stfem_mesh_vertices(ncell, lower, upper, factor, seed, 0, ncell[2], vertices.data()),
          "stfem_mesh_vertices");
  }
};

// include/operators.h:967-1191.  n_components must be 1, dim must be 3 in this round.
template <int dim, int n_components, typename Number> class MatrixFreeOperator {
  static_assert(dim == 3 && n_components == 1, "scalar 3D path");

public:
  using VectorType = VectorT<Number>;
  using BlockVectorType = BlockVectorT<Number>;

  // reference: (mapping, dof_handler, constraints, quadrature, mass_scaling, laplace_scaling);
  // here the mesh + degree stand for mapping/dof_handler/constraints/QGauss(degree+1)
  MatrixFreeOperator(const Mesh &mesh, unsigned fe_degree, double mass_matrix_scaling, double laplace_matrix_scaling)
    : mass_matrix_scaling(mass_matrix_scaling), laplace_matrix_scaling(laplace_matrix_scaling)
  {
    ctx_ = make_context(mesh, fe_degree);
  }
  // K and M of one SystemMatrix share the MatrixFree data: build the second from the first
  MatrixFreeOperator(const MatrixFreeOperator &other, double mass_matrix_scaling, double laplace_matrix_scaling)
    : mass_matrix_scaling(mass_matrix_scaling), laplace_matrix_scaling(laplace_matrix_scaling), ctx_(other.ctx_)
  {}

  template <typename Number2> void initialize_dof_vector(VectorT<Number2> &vec) const { vec.reinit(ctx_); }

  // z-slab partition of the mesh (parallel::distributed::Triangulation in the reference, tests/tp_01.cc:80):
  // this rank's mesh is one slab (Mesh::dirichlet_mask without the interface faces); every vmult then
  // brackets its cell loop as MatrixFree::cell_loop does (operators.h:1016-1017): ghost update of src,
  // sweep, one packed add-exchange of the interface planes of dst.  Shared by all operators on this context.
  void set_partition(const std::shared_ptr<Communicator> &comm, int lower_rank, int upper_rank)
  {
    ctx_->comm = comm;
    ctx_->lower_rank = lower_rank;
    ctx_->upper_rank = upper_rank;
  }

  // General (perturbed) meshes on a partition: the mesh of this slab plus ONE ghost cell layer on every side with a neighbour rank
  // (vertices of the neighbour's first cell layer; Mesh::dirichlet_mask as the slab's).  The reference's smoother builds its cell
  // blocks on locally owned and ghost cells (stmg.h:688-689, 795-796); PreconditionVanka does the same from this mesh.
  void set_ghost_layers(const Mesh &extended_mesh) { ctx_->extended = make_context(extended_mesh, ctx_->degree); }

  void vmult(VectorType &dst, const VectorType &src, void *stream = nullptr) const
  {
    ghost_update(*ctx_, src.handle(), stream);
    check(stfem_space_vmult(ctx_->h, mass_matrix_scaling, laplace_matrix_scaling, dst.handle(), src.handle(), stream),
          "MatrixFreeOperator::vmult");
    compress_add(*ctx_, dst.handle(), stream);
  }

  // operators.h:1060-1087; one value per cell, or per (cell, quadrature point)
  // Both coefficients are filled where both scalings are nonzero (1071-1085).  NB: the coefficient tables
  // live on the shared Context (one MatrixFree), not per operator as in the reference: K and M built on
  // one context see each other's coefficients, which is what tests/tp_01.cc:141-150 sets up anyway
  // (coefficient on K only, whose mass scaling is 0).
  void evaluate_coefficient(const std::vector<double> &values)
  {
    const size_t ncells = size_t(stfem_n_cells(ctx_->h));
    const int layout = values.size() == ncells ? 1 : 2;
    if (mass_matrix_scaling != 0.0) check(stfem_set_coefficient(ctx_->h, 0, layout, values.data()), "evaluate_coefficient");
    if (laplace_matrix_scaling != 0.0)
      check(stfem_set_coefficient(ctx_->h, 1, layout, values.data()), "evaluate_coefficient");
  }

  unsigned long long m() const { return (unsigned long long)stfem_n_dofs(ctx_->h); }
  Number el(unsigned, unsigned) const { throw std::logic_error("MatrixFreeOperator::el is not implemented"); }

  VectorType get_matrix_diagonal(void *stream = nullptr) const
  {
    VectorType d;
    d.reinit(ctx_);
    check(stfem_diagonal(ctx_->h, mass_matrix_scaling, laplace_matrix_scaling, d.handle(), stream),
          "get_matrix_diagonal");
    return d;
  }

  // operators.h:1041-1045, 1106-1109: 1 / d where |d| > sqrt(eps), 1 elsewhere
  VectorType get_matrix_diagonal_inverse(void *stream = nullptr) const
  {
    VectorType d;
    d.reinit(ctx_);
    check(stfem_diagonal_inverse(ctx_->h, mass_matrix_scaling, laplace_matrix_scaling, d.handle(), stream),
          "get_matrix_diagonal_inverse");
    return d;
  }

  const std::shared_ptr<Context> &context() const { return ctx_; }
  const double mass_matrix_scaling, laplace_matrix_scaling;

private:
  static std::shared_ptr<Context> make_context(const Mesh &mesh, unsigned degree)
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
    stfem_space_desc sd{int32_t(degree), int32_t(degree + 1), 1, std::is_same<Number, float>::value ? 1 : 0};
    stfem_ctx *c = nullptr;
    check(stfem_ctx_create(&md, &sd, &c), "stfem_ctx_create");
    auto ctx = std::make_shared<Context>(c);
    ctx->degree = degree;
    return ctx;
  }
  std::shared_ptr<Context> ctx_;
};
template <int dim, typename Number> using MatrixFreeOperatorScalar = MatrixFreeOperator<dim, 1, Number>;

// include/operators.h:328-663 (SystemMatrixBase + SystemMatrix): A = Alpha (x) K + Beta (x) M
template <int dim, typename Number, typename SystemMatrixTypeK, typename SystemMatrixTypeM = SystemMatrixTypeK>
class SystemMatrix {
public:
  using BlockVectorType = BlockVectorT<Number>;
  using VectorType = VectorT<Number>;

  // The reference keeps references to K, M, Alpha, Beta (operators.h:465-469); callers keep them alive.
  SystemMatrix(const SystemMatrixTypeK &K, const SystemMatrixTypeM &M, const FullMatrix<Number> &Alpha_,
               const FullMatrix<Number> &Beta_)
    : K(K), M(M), Alpha(Alpha_), Beta(Beta_), alpha_is_zero(Alpha_.all_zero()), beta_is_zero(Beta_.all_zero())
  {
    if (Alpha.m() != Beta.m() || Alpha.n() != Beta.n()) throw std::invalid_argument("Alpha/Beta shape mismatch");
    if (K.context() != M.context()) throw std::invalid_argument("K and M must share one MatrixFree context");
    if (K.laplace_matrix_scaling != 1.0 || K.mass_matrix_scaling != 0.0 || M.mass_matrix_scaling != 1.0 ||
        M.laplace_matrix_scaling != 0.0)
      throw std::invalid_argument("fused path expects K = (0,1) and M = (1,0) as in tests/tp_01.cc:114-117");
  }
  virtual ~SystemMatrix() = default;

  virtual void vmult(BlockVectorType &dst, const BlockVectorType &src, void *stream = nullptr) const
  {
    apply(dst, src, 0, 0, stream);
  }
  virtual void Tvmult(BlockVectorType &dst, const BlockVectorType &src, void *stream = nullptr) const
  {
    apply(dst, src, 1, 0, stream);
  }
  // n x 1 case for rhs assembly (operators.h:377-382, 586-611)
  virtual void vmult_slice_add(BlockVectorType &dst, const BlockVectorType &src, void *stream = nullptr) const
  {
    apply(dst, src, 0, 1, stream);
  }
  void vmult_slice(BlockVectorType &dst, const BlockVectorType &src, void *stream = nullptr) const
  {
    apply(dst, src, 0, 0, stream);
  }
  virtual void form(BlockVectorType &dst, const BlockVectorType &src) const { vmult(dst, src); }

  unsigned long long m() const { return Alpha.m() * M.m(); }
  unsigned long long n() const { return m(); }
  Number el(unsigned, unsigned) const { throw std::logic_error("SystemMatrix::el is not implemented"); }

  template <typename Number2> void initialize_dof_vector(VectorT<Number2> &vec, unsigned = 1) const
  {
    vec.reinit(K.context());
  }
  template <typename Number2> void initialize_dof_vector(BlockVectorT<Number2> &vec) const
  {
    vec.reinit(K.context(), Alpha.m());
  }

  // operators.h:613-623: block i = Alpha(i,i) diag K + Beta(i,i) diag M
  BlockVectorType get_matrix_diagonal(void *stream = nullptr) const { return diagonal(0, stream); }
  // operators.h:625-637 (as the reference combines it): 1/Alpha(i,i) (diag K)^-1 + 1/Beta(i,i) (diag M)^-1
  BlockVectorType get_matrix_diagonal_inverse(void *stream = nullptr) const { return diagonal(1, stream); }
  // operators.h:384-388: linear operator, nothing to linearise around
  virtual void set_data(const BlockVectorType &) const {}

private:
  BlockVectorType diagonal(int inverse, void *stream) const
  {
    if (Alpha.m() != Alpha.n()) throw std::invalid_argument("diagonal of a non-square system");
    BlockVectorType d;
    d.reinit(K.context(), Alpha.m());
    std::vector<double> a(size_t(Alpha.m()) * Alpha.n()), b(a.size());
    for (size_t i = 0; i < a.size(); ++i) {
      a[i] = double(Alpha.data()[i]);
      b[i] = double(Beta.data()[i]);
    }
    check(stfem_st_diagonal(K.context()->h, int(Alpha.m()), a.data(), b.data(), inverse, d.handle(), stream),
          "SystemMatrix::get_matrix_diagonal");
    return d;
  }
  void apply(BlockVectorType &dst, const BlockVectorType &src, int transpose, int add, void *stream) const
  {
    // the C-ABI takes the temporal matrices in double whatever the operator's Number is
    std::vector<double> a(size_t(Alpha.m()) * Alpha.n()), b(a.size());
    for (size_t i = 0; i < a.size(); ++i) {
      a[i] = double(Alpha.data()[i]);
      b[i] = double(Beta.data()[i]);
    }
    const Context &c = *K.context();
    ghost_update(c, src.handle(), stream);
    if (add && c.partitioned()) { // the exchange adds the partials of THIS product only
      BlockVectorType tmp;
      tmp.reinit(K.context(), dst.n_blocks());
      check(stfem_st_vmult(c.h, int(Alpha.m()), int(Alpha.n()), a.data(), b.data(), transpose, 0, tmp.handle(),
                           src.handle(), stream),
            "SystemMatrix::vmult");
      compress_add(c, tmp.handle(), stream);
      std::vector<double> eye(size_t(dst.n_blocks()) * dst.n_blocks(), 0.0);
      for (unsigned i = 0; i < dst.n_blocks(); ++i) eye[size_t(i) * dst.n_blocks() + i] = 1.0;
      check(stfem_tensorproduct_add(c.h, int(dst.n_blocks()), int(dst.n_blocks()), eye.data(), dst.handle(), tmp.handle(),
                                    stream),
            "SystemMatrix::vmult_slice_add");
      return;
    }
    check(stfem_st_vmult(c.h, int(Alpha.m()), int(Alpha.n()), a.data(), b.data(), transpose, add,
                         dst.handle(), src.handle(), stream),
          "SystemMatrix::vmult");
    compress_add(c, dst.handle(), stream);
  }
  const SystemMatrixTypeK &K;
  const SystemMatrixTypeM &M;
  const FullMatrix<Number> &Alpha;
  const FullMatrix<Number> &Beta;
  bool alpha_is_zero, beta_is_zero;
};

// include/stmg.h:619-907 PreconditionVanka: cell-patch additive-Schwarz smoother of Alpha (x) K + Beta (x) M.
// The reference builds it from the assembled sparse matrices K_, M_ (tests/tp_01.cc:283-321); here the blocks
// come from the context's mesh and degree directly (what those matrices are assembled from), so the constructor
// takes the operator instead.  vmult overwrites dst; smooth = vmult (stmg.h:881-885).
template <typename Number> class PreconditionVanka {
public:
  using BlockVectorType = BlockVectorT<Number>;
  template <typename OperatorType>
  PreconditionVanka(const OperatorType &K, const FullMatrix<Number> &Alpha, const FullMatrix<Number> &Beta) : ctx_(K.context())
  {
    if (Alpha.m() != Alpha.n() || Beta.m() != Alpha.m() || Beta.n() != Alpha.n()) throw std::invalid_argument("Alpha/Beta must be square and of one size");
    std::vector<double> a(size_t(Alpha.m()) * Alpha.n()), b(a.size());
    for (size_t i = 0; i < a.size(); ++i) {
      a[i] = double(Alpha.data()[i]);
      b[i] = double(Beta.data()[i]);
    }
    stfem_vanka *v = nullptr;
    // on a slab of a partitioned mesh (MatrixFreeOperator::set_partition BEFORE this constructor) the cells behind the interface faces count
    const int neighbours = (ctx_->lower_rank >= 0 ? 16 : 0) | (ctx_->upper_rank >= 0 ? 32 : 0);
    const int rc = (neighbours && ctx_->extended)
                     ? stfem_vanka_create_partitioned_general(ctx_->h, ctx_->extended->h, int(Alpha.m()), a.data(), b.data(), neighbours, &v)
                     : stfem_vanka_create_partitioned(ctx_->h, int(Alpha.m()), a.data(), b.data(), neighbours, &v);
    if (rc != STFEM_OK) throw Error(rc, std::string("stfem_vanka_create: ") + stfem_vanka_last_error());
    v_.reset(v, stfem_vanka_destroy);
  }
  void vmult(BlockVectorType &dst, const BlockVectorType &src, void *stream = nullptr) const
  {
    const int rc = stfem_vanka_vmult(v_.get(), dst.handle(), src.handle(), stream);
    if (rc != STFEM_OK) throw Error(rc, std::string("PreconditionVanka::vmult: ") + stfem_vanka_last_error());
    compress_add(*ctx_, dst.handle(), stream); // partitioned: the interface planes hold partial sums (dst.compress(add) in the reference)
  }
  // dst = (accumulate ? dst : 0) + omega * vmult(src): the step of PreconditionRelaxation (stmg.h:1199-1238), fused into the
  // smoother's scatter (stfem_vanka_step).  On a slab of a partitioned mesh the partial sums of the interface planes have to be
  // completed before they may be added to dst: there the update stays a pass of its own.
  void step(BlockVectorType &dst, double omega, bool accumulate, const BlockVectorType &src, void *stream = nullptr) const
  {
    const bool partitioned = ctx_->lower_rank >= 0 || ctx_->upper_rank >= 0;
    if (partitioned && accumulate) {
      if (!tmp_.handle() || tmp_.n_blocks() != dst.n_blocks()) tmp_.reinit(ctx_, dst.n_blocks());
      vmult(tmp_, src, stream);
      check(stfem_vector_axpby(ctx_->h, omega, tmp_.handle(), 1.0, dst.handle(), stream), "stfem_vector_axpby");
      return;
    }
    const int rc = stfem_vanka_step(v_.get(), dst.handle(), omega, accumulate ? 1 : 0, src.handle(), stream);
    if (rc != STFEM_OK) throw Error(rc, std::string("PreconditionVanka::step: ") + stfem_vanka_last_error());
    compress_add(*ctx_, dst.handle(), stream);
  }
  void smooth(BlockVectorType &u, const BlockVectorType &rhs) const { vmult(u, rhs); }
  void clear() { v_.reset(); }
  int n_classes() const { return stfem_vanka_n_classes(v_.get()); }

private:
  std::shared_ptr<Context> ctx_;
  std::shared_ptr<stfem_vanka> v_;
  mutable BlockVectorType tmp_;
};

// include/operators.h:1953-2050 PDE<>: the nonlinear-solver face of an operator.  residual = rhs - form(src);
// form falls back to vmult for operators without one (internal::has_form, 1999-2004); vmult applies the
// Jacobian operator (the same object unless given separately).
template <int dim, typename Number, typename PDEOperator, typename JacOperator = PDEOperator> class PDE {
  using BlockVectorType = BlockVectorT<Number>;

  template <typename Op, typename = void> struct has_form : std::false_type {};
  template <typename Op>
  struct has_form<Op, std::void_t<decltype(std::declval<const Op &>().form(std::declval<BlockVectorType &>(),
                                                                          std::declval<const BlockVectorType &>()))>>
    : std::true_type {};

public:
  void init(const PDEOperator &pde_operator_, const BlockVectorType &rhs_)
  {
    pde_operator = &pde_operator_;
    jac_operator = &pde_operator_;
    rhs = &rhs_;
  }
  void init(const PDEOperator &pde_operator_, const JacOperator &jac_operator_, const BlockVectorType &rhs_)
  {
    pde_operator = &pde_operator_;
    jac_operator = &jac_operator_;
    rhs = &rhs_;
  }
  void set_rhs(const BlockVectorType &rhs_) const { rhs = &rhs_; }
  void set_data(const BlockVectorType &data) const
  {
    pde_operator->set_data(data);
    if (static_cast<const void *>(pde_operator) != static_cast<const void *>(jac_operator)) jac_operator->set_data(data);
  }
  void residual(BlockVectorType &dst, const BlockVectorType &src, const BlockVectorType &rhs_) const
  {
    rhs = &rhs_;
    residual(dst, src);
  }
  // rhs - form(src): dst = -form(src) + rhs through the block BLAS-1 of the boundary
  void residual(BlockVectorType &dst, const BlockVectorType &src) const
  {
    form(dst, src);
    const unsigned n = dst.n_blocks();
    FullMatrix<Number> minus_two(n, n), one(n, n);
    for (unsigned i = 0; i < n; ++i) {
      minus_two(i, i) = Number(-2); // dst += -2 dst  ->  -form
      one(i, i) = Number(1);
    }
    BlockVectorType tmp;
    tmp.reinit(dst.context(), n);
    tensorproduct_add_impl(dst.context(), tmp, one, dst);       // tmp = form
    tensorproduct_add_impl(dst.context(), dst, minus_two, tmp); // dst = form - 2 form
    tensorproduct_add_impl(dst.context(), dst, one, *rhs);      // dst = rhs - form
  }
  void form(BlockVectorType &dst, const BlockVectorType &src) const
  {
    if constexpr (has_form<PDEOperator>::value) pde_operator->form(dst, src);
    else pde_operator->vmult(dst, src);
  }
  void vmult(BlockVectorType &dst, const BlockVectorType &src) const { jac_operator->vmult(dst, src); }
  template <typename Number2> void initialize_dof_vector(VectorT<Number2> &vec, unsigned i = 0) const
  {
    pde_operator->initialize_dof_vector(vec, i);
  }
  template <typename Number2> void initialize_dof_vector(BlockVectorT<Number2> &vec) const
  {
    pde_operator->initialize_dof_vector(vec);
  }

private:
  static void tensorproduct_add_impl(const std::shared_ptr<Context> &ctx, BlockVectorType &c, const FullMatrix<Number> &A,
                                     const BlockVectorType &b)
  {
    std::vector<double> a(size_t(A.m()) * A.n());
    for (size_t i = 0; i < a.size(); ++i) a[i] = double(A.data()[i]);
    check(stfem_tensorproduct_add(ctx->h, int(A.m()), int(A.n()), a.data(), c.handle(), b.handle(), nullptr),
          "PDE::residual");
  }
  mutable const BlockVectorType *rhs = nullptr;
  const PDEOperator *pde_operator = nullptr;
  const JacOperator *jac_operator = nullptr;
};

// operators.h:211-283 tensorproduct_add: c_i += A(i,j) b_j
template <typename Number>
void tensorproduct_add(const std::shared_ptr<Context> &ctx, BlockVectorT<Number> &c, const FullMatrix<Number> &A,
                       const BlockVectorT<Number> &b, void *stream = nullptr)
{
  std::vector<double> a(size_t(A.m()) * A.n());
  for (size_t i = 0; i < a.size(); ++i) a[i] = double(A.data()[i]);
  check(stfem_tensorproduct_add(ctx->h, int(A.m()), int(A.n()), a.data(), c.handle(), b.handle(), stream),
        "tensorproduct_add");
}

} // namespace stfem
