// Host-side mirror of the reference's vector types (include/types.h:19-36) on top of the C-ABI.
// VectorT<Number>  ~ dealii::LinearAlgebra::distributed::Vector<Number>      (one spatial vector)
// BlockVectorT     ~ dealii::LinearAlgebra::distributed::BlockVector<Number> (one per temporal DoF)
// Number = double (solver) and Number = float (multigrid levels) are both built on the device;
// host-side copies always go through double (the C-ABI converts for fp32 contexts).
#pragma once
#include "../../../include/stfem.h"

#include <array>
#include <cmath>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace stfem {

// a named trace range for the time of a scope (roctx through the C-ABI): the reference's TimerOutput::Scope
struct TraceRange {
  explicit TraceRange(const char *name) { stfem_trace_push(name); }
  ~TraceRange() { stfem_trace_pop(); }
  TraceRange(const TraceRange &) = delete;
  TraceRange &operator=(const TraceRange &) = delete;
};


struct Error : std::runtime_error {
  int status;
  Error(int s, const std::string &what)
    : std::runtime_error(what + ": " + stfem_strerror(s) + " [" + stfem_last_hip_error() + "]"), status(s) {}
};
inline void check(int status, const char *what)
{
  if (status != STFEM_OK) throw Error(status, what);
}

// minimal stand-in for dealii::FullMatrix<Number> (row-major)
template <typename Number> class FullMatrix {
public:
  FullMatrix() = default;
  FullMatrix(unsigned m, unsigned n) : m_(m), n_(n), v_(size_t(m) * n, Number(0)) {}
  unsigned m() const { return m_; }
  unsigned n() const { return n_; }
  Number &operator()(unsigned i, unsigned j) { return v_[size_t(i) * n_ + j]; }
  const Number &operator()(unsigned i, unsigned j) const { return v_[size_t(i) * n_ + j]; }
  bool all_zero() const
  {
    for (const Number &x : v_)
      if (x != Number(0)) return false;
    return true;
  }
  const Number *data() const { return v_.data(); }
  Number *data() { return v_.data(); }
  // C = this * B
  void mmult(FullMatrix &C, const FullMatrix &B) const
  {
    C = FullMatrix(m_, B.n_);
    for (unsigned i = 0; i < m_; ++i)
      for (unsigned k = 0; k < n_; ++k)
        for (unsigned j = 0; j < B.n_; ++j) C(i, j) += (*this)(i, k) * B(k, j);
  }
  FullMatrix &operator*=(Number s)
  {
    for (Number &x : v_) x *= s;
    return *this;
  }
  // in-place inverse (dealii::FullMatrix::gauss_jordan), partial pivoting
  void gauss_jordan()
  {
    if (m_ != n_) throw std::invalid_argument("gauss_jordan: square matrices only");
    const unsigned n = n_;
    std::vector<double> A(v_.begin(), v_.end()), I(size_t(n) * n, 0.0);
    for (unsigned i = 0; i < n; ++i) I[size_t(i) * n + i] = 1.0;
    for (unsigned c = 0; c < n; ++c) {
      unsigned p = c;
      for (unsigned r = c + 1; r < n; ++r)
        if (std::abs(A[size_t(r) * n + c]) > std::abs(A[size_t(p) * n + c])) p = r;
      if (A[size_t(p) * n + c] == 0.0) throw std::runtime_error("gauss_jordan: singular matrix");
      for (unsigned k = 0; k < n; ++k) {
        std::swap(A[size_t(c) * n + k], A[size_t(p) * n + k]);
        std::swap(I[size_t(c) * n + k], I[size_t(p) * n + k]);
      }
      const double inv = 1.0 / A[size_t(c) * n + c];
      for (unsigned k = 0; k < n; ++k) {
        A[size_t(c) * n + k] *= inv;
        I[size_t(c) * n + k] *= inv;
      }
      for (unsigned r = 0; r < n; ++r) {
        if (r == c) continue;
        const double f = A[size_t(r) * n + c];
        for (unsigned k = 0; k < n; ++k) {
          A[size_t(r) * n + k] -= f * A[size_t(c) * n + k];
          I[size_t(r) * n + k] -= f * I[size_t(c) * n + k];
        }
      }
    }
    for (size_t i = 0; i < v_.size(); ++i) v_[i] = Number(I[i]);
  }
  template <typename N2> FullMatrix<N2> cast() const
  {
    FullMatrix<N2> r(m_, n_);
    for (size_t i = 0; i < v_.size(); ++i) r.data()[i] = N2(v_[i]);
    return r;
  }

private:
  unsigned m_ = 0, n_ = 0;
  std::vector<Number> v_;
};

// The MPI side of LinearAlgebra::distributed::Vector for z-slab partitions: one RCCL communicator per
// process (include/stfem.h: stfem_comm_*).  The 128-byte id comes from unique_id() on one rank and is
// broadcast by the caller (MPI_Bcast in a deal.II program).
class Communicator {
public:
  using UniqueId = std::array<char, STFEM_COMM_ID_BYTES>;
  static UniqueId unique_id()
  {
    UniqueId id;
    check_comm(stfem_comm_get_unique_id(id.data()), "stfem_comm_get_unique_id");
    return id;
  }
  Communicator(const UniqueId &id, int rank, int world, int device)
  {
    check_comm(stfem_comm_create(id.data(), rank, world, device, &h_), "stfem_comm_create");
  }
  ~Communicator() { stfem_comm_destroy(h_); }
  Communicator(const Communicator &) = delete;
  Communicator &operator=(const Communicator &) = delete;
  int rank() const { return stfem_comm_rank(h_); }
  int size() const { return stfem_comm_size(h_); }
  stfem_comm *handle() const { return h_; }
  static void check_comm(int status, const char *what)
  {
    if (status == STFEM_ERR_COMM || status == STFEM_ERR_UNSUPPORTED)
      throw Error(status, std::string(what) + ": " + stfem_comm_last_error());
    check(status, what);
  }

private:
  stfem_comm *h_ = nullptr;
};

// owning handle of an stfem_ctx, shared by the K and M operators of one mesh; on a partitioned mesh
// also the communicator and the ranks holding the slabs below / above (-1: none)
struct Context {
  stfem_ctx *h = nullptr;
  std::shared_ptr<Communicator> comm;
  int lower_rank = -1, upper_rank = -1;
  unsigned degree = 0; // FE_Q(degree) in space
  // general meshes on a partition: the same slab plus one ghost cell layer per interface (MatrixFreeOperator::set_ghost_layers):
  // what the cell-patch smoother builds its blocks on (stmg.h:688-689: locally owned and ghost cells)
  std::shared_ptr<Context> extended;
  bool owned = true; // false: the handle belongs to another object (the pressure space of a Stokes context)
  explicit Context(stfem_ctx *c, bool owned = true) : h(c), owned(owned) {}
  ~Context() { if (owned) stfem_ctx_destroy(h); }
  Context(const Context &) = delete;
  Context &operator=(const Context &) = delete;
  bool partitioned() const { return comm && (lower_rank >= 0 || upper_rank >= 0); }
  // entries of a block this rank owns: all but the top plane where a slab above owns it
  int64_t n_owned() const
  {
    int32_t nd[3];
    check(stfem_n_dofs_1d(h, nd), "stfem_n_dofs_1d");
    return int64_t(nd[0]) * nd[1] * (upper_rank >= 0 ? nd[2] - 1 : nd[2]);
  }
};

template <typename Number> class BlockVectorT;

template <typename Number> class VectorT {

public:
  VectorT() = default;
  void reinit(const std::shared_ptr<Context> &ctx)
  {
    ctx_ = ctx;
    stfem_vec *v = nullptr;
    check(stfem_vector_create(ctx->h, 1, &v), "stfem_vector_create");
    v_.reset(v, stfem_vector_destroy);
  }
  size_t size() const { return ctx_ ? size_t(stfem_n_dofs(ctx_->h)) : 0; }
  void copy_from_host(const std::vector<double> &h)
  {
    const double *p[1] = {h.data()};
    check(stfem_vector_upload(v_.get(), p), "stfem_vector_upload");
  }
  std::vector<double> copy_to_host() const
  {
    std::vector<double> h(size());
    double *p[1] = {h.data()};
    check(stfem_vector_download(v_.get(), p), "stfem_vector_download");
    return h;
  }
  stfem_vec *handle() const { return v_.get(); }
  const std::shared_ptr<Context> &context() const { return ctx_; }

private:
  std::shared_ptr<Context> ctx_;
  std::shared_ptr<stfem_vec> v_;
};

template <typename Number> class BlockVectorT {
public:
  BlockVectorT() = default;
  void reinit(const std::shared_ptr<Context> &ctx, unsigned n_blocks)
  {
    ctx_ = ctx;
    nb_ = n_blocks;
    stfem_vec *v = nullptr;
    check(stfem_vector_create(ctx->h, int(n_blocks), &v), "stfem_vector_create");
    v_.reset(v, stfem_vector_destroy);
  }
  // view of caller-owned device arrays (one pointer per block); nothing is copied or freed
  void wrap(const std::shared_ptr<Context> &ctx, void *const *device_blocks, unsigned n_blocks)
  {
    ctx_ = ctx;
    nb_ = n_blocks;
    stfem_vec *v = nullptr;
    check(stfem_vector_wrap(ctx->h, int(n_blocks), device_blocks, &v), "stfem_vector_wrap");
    v_.reset(v, stfem_vector_destroy);
  }
  unsigned n_blocks() const { return nb_; }
  size_t block_size() const { return ctx_ ? size_t(stfem_n_dofs(ctx_->h)) : 0; }
  void copy_from_host(const std::vector<std::vector<double>> &h)
  {
    std::vector<const double *> p;
    for (const auto &b : h) p.push_back(b.data());
    check(stfem_vector_upload(v_.get(), p.data()), "stfem_vector_upload");
  }
  std::vector<std::vector<double>> copy_to_host() const
  {
    std::vector<std::vector<double>> h(nb_, std::vector<double>(block_size()));
    std::vector<double *> p;
    for (auto &b : h) p.push_back(b.data());
    check(stfem_vector_download(v_.get(), p.data()), "stfem_vector_download");
    return h;
  }
  stfem_vec *handle() const { return v_.get(); }
  const std::shared_ptr<Context> &context() const { return ctx_; }

private:
  std::shared_ptr<Context> ctx_;
  std::shared_ptr<stfem_vec> v_;
  unsigned nb_ = 0;
};

} // namespace stfem
