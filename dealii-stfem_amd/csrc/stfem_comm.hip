// RCCL inside the boundary: the z-slab partition's interface-plane exchange and the
// reducing inner product, behind the C-ABI, so that a C++ caller (the deal.II side) runs sharded
// without any Python.
//
// Replaces, around the cell loop of MatrixFreeOperator::vmult (reference include/operators.h:1016-1017,
// MatrixFree::cell_loop): src.update_ghost_values() -> stfem_ghost_update, dst.compress(VectorOperation::add)
// followed by the next update_ghost_values -> stfem_halo_begin / stfem_halo_end (one packed exchange of
// all temporal blocks per space-time vmult instead of the reference's 2 * n_blocks), and the MPI_Allreduce
// inside LinearAlgebra::distributed::Vector::operator* / l2_norm -> stfem_dot_global.
//
// One process per GPU; a communicator is one RCCL communicator + one private HIP stream + two events.
// Ring neighbours only (z-slabs): each message rides one xGMI link; the four transfers of a rank are
// fused in one ncclGroupStart/End.  RCCL is bound at run time (dlopen): the library also loads on a
// box without RCCL, where stfem_comm_create reports STFEM_ERR_UNSUPPORTED.  If the process already
// carries an RCCL (e.g. PyTorch's) that copy is used, so that one process never runs two of them.
#include "../../include/stfem.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h> // types and enums only: no link-time dependency

#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <new>

namespace {

struct Rccl {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr; // optional
  bool ok = false;
};

thread_local char g_comm_err[256] = "";

template <class F> bool bind(void *h, const char *name, F &f)
{
  f = reinterpret_cast<F>(dlsym(h, name));
  return f != nullptr;
}

const Rccl &rccl()
{
  static Rccl r = [] {
    Rccl q;
    // a copy already in the process first (RTLD_DEFAULT sees it if it was loaded globally), then
    // the known file names without loading a second copy, then a fresh load
    void *h = nullptr;
    if (dlsym(RTLD_DEFAULT, "ncclCommInitRank")) h = RTLD_DEFAULT;
    const char *names[] = {"librccl.so", "librccl.so.1"};
    for (int pass = 0; pass < 2 && !h; ++pass)
      for (const char *n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
        if (h) break;
      }
    if (!h) return q;
    q.handle = h;
    q.ok = bind(h, "ncclGetUniqueId", q.GetUniqueId) && bind(h, "ncclCommInitRank", q.CommInitRank) &&
           bind(h, "ncclCommDestroy", q.CommDestroy) && bind(h, "ncclSend", q.Send) && bind(h, "ncclRecv", q.Recv) &&
           bind(h, "ncclGroupStart", q.GroupStart) && bind(h, "ncclGroupEnd", q.GroupEnd) &&
           bind(h, "ncclAllReduce", q.AllReduce) && bind(h, "ncclGetErrorString", q.GetErrorString);
    (void)bind(h, "ncclCommCount", q.CommCount);
    return q;
  }();
  return r;
}

} // namespace

struct stfem_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  hipStream_t stream = nullptr;     // the transfers run here, beside the caller's stream
  hipEvent_t packed = nullptr, arrived = nullptr;
  // packed planes [top send | bottom send | top recv | bottom recv], grown on demand
  void *buf = nullptr;
  size_t buf_bytes = 0;
  double *d_red = nullptr; // all-reduce operand
  // the exchange in flight between halo_begin and halo_end
  bool pending = false;
  int lower = -1, upper = -1;
  size_t plane_bytes = 0; // of all blocks
};

#define COMM_HIP(call)                                                                    \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) {                                                               \
      snprintf(g_comm_err, sizeof(g_comm_err), "%s: %s", #call, hipGetErrorString(e_));   \
      return STFEM_ERR_HIP;                                                               \
    }                                                                                     \
  } while (0)
#define COMM_NCCL(call)                                                                   \
  do {                                                                                    \
    ncclResult_t r_ = (call);                                                             \
    if (r_ != ncclSuccess) {                                                              \
      snprintf(g_comm_err, sizeof(g_comm_err), "%s: %s", #call, rccl().GetErrorString(r_)); \
      return STFEM_ERR_COMM;                                                              \
    }                                                                                     \
  } while (0)

// inside ncclGroupStart / GroupEnd: a failed call still closes the group (an open group would swallow every later
// RCCL call of this thread) before the error is returned
#define COMM_NCCL_IN_GROUP(call)                                                          \
  do {                                                                                    \
    ncclResult_t r_ = (call);                                                             \
    if (r_ != ncclSuccess) {                                                              \
      snprintf(g_comm_err, sizeof(g_comm_err), "%s: %s", #call, rccl().GetErrorString(r_)); \
      (void)rccl().GroupEnd();                                                            \
      return STFEM_ERR_COMM;                                                              \
    }                                                                                     \
  } while (0)

extern "C" {

const char *stfem_comm_last_error(void) { return g_comm_err; }

int stfem_comm_get_unique_id(void *id)
{
  if (!id) return STFEM_ERR_INVALID_ARGUMENT;
  static_assert(STFEM_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
  if (!rccl().ok) {
    const char *why = dlerror(); // (a second call returns NULL)
    snprintf(g_comm_err, sizeof(g_comm_err), "librccl.so not found: %s", why ? why : "");
    return STFEM_ERR_UNSUPPORTED;
  }
  ncclUniqueId u;
  COMM_NCCL(rccl().GetUniqueId(&u));
  std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
  return STFEM_OK;
}

int stfem_comm_create(const void *id, int rank, int world, int device, stfem_comm **out)
{
  if (!id || !out || world < 1 || rank < 0 || rank >= world) return STFEM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (!rccl().ok) {
    snprintf(g_comm_err, sizeof(g_comm_err), "librccl.so not found");
    return STFEM_ERR_UNSUPPORTED;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return STFEM_ERR_NO_DEVICE;
  if (device < 0 || device >= ndev) return STFEM_ERR_INVALID_ARGUMENT;
  COMM_HIP(hipSetDevice(device));
  stfem_comm *c = new (std::nothrow) stfem_comm;
  if (!c) return STFEM_ERR_OUT_OF_MEMORY;
  c->rank = rank; c->world = world; c->device = device;
  ncclUniqueId u;
  std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
  ncclResult_t r = rccl().CommInitRank(&c->comm, world, u, rank);
  if (r != ncclSuccess) {
    snprintf(g_comm_err, sizeof(g_comm_err), "ncclCommInitRank: %s", rccl().GetErrorString(r));
    delete c;
    return STFEM_ERR_COMM;
  }
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->packed, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->arrived, hipEventDisableTiming) != hipSuccess ||
      hipMalloc(&c->d_red, sizeof(double)) != hipSuccess) {
    stfem_comm_destroy(c);
    return STFEM_ERR_HIP;
  }
  *out = c;
  return STFEM_OK;
}

void stfem_comm_destroy(stfem_comm *c)
{
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)rccl().CommDestroy(c->comm);
  if (c->buf) (void)hipFree(c->buf);
  if (c->d_red) (void)hipFree(c->d_red);
  if (c->packed) (void)hipEventDestroy(c->packed);
  if (c->arrived) (void)hipEventDestroy(c->arrived);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int stfem_comm_available(void) { return rccl().ok ? 1 : 0; }
int stfem_comm_rccl_count(const stfem_comm *c)
{
  int n = 0;
  if (!c || !c->comm || !rccl().CommCount || rccl().CommCount(c->comm, &n) != ncclSuccess) return 0;
  return n;
}
int stfem_comm_rank(const stfem_comm *c) { return c ? c->rank : -1; }
int stfem_comm_size(const stfem_comm *c) { return c ? c->world : 0; }

static int plane_geometry(stfem_ctx *ctx, const stfem_vec *v, size_t &plane_bytes, int &nz)
{
  int32_t nd[3];
  if (stfem_n_dofs_1d(ctx, nd) != STFEM_OK) return STFEM_ERR_INVALID_ARGUMENT;
  const size_t es = stfem_ctx_precision(ctx) ? sizeof(float) : sizeof(double);
  plane_bytes = size_t(nd[0]) * nd[1] * es * stfem_vector_n_blocks(v);
  nz = nd[2];
  return STFEM_OK;
}

static int ensure_buffers(stfem_comm *c, size_t plane_bytes)
{
  if (c->buf_bytes >= 4 * plane_bytes) return STFEM_OK;
  COMM_HIP(hipStreamSynchronize(c->stream));
  if (c->buf) COMM_HIP(hipFree(c->buf));
  c->buf = nullptr;
  c->buf_bytes = 0;
  if (hipMalloc(&c->buf, 4 * plane_bytes) != hipSuccess) return STFEM_ERR_OUT_OF_MEMORY;
  c->buf_bytes = 4 * plane_bytes;
  return STFEM_OK;
}

static int check_neighbours(const stfem_comm *c, int lower, int upper)
{
  return (lower >= -1 && lower < c->world && upper >= -1 && upper < c->world) ? STFEM_OK : STFEM_ERR_INVALID_ARGUMENT;
}

int stfem_halo_begin(stfem_ctx *ctx, stfem_comm *c, stfem_vec *v, int lower, int upper, void *stream)
{
  return stfem_halo_begin_split(c, ctx, v, ctx, v, lower, upper, stream);
}

// The same with the two interface planes taken from two vectors: plane 0 of v_lo (context ctx_lo) goes to the lower neighbour, the
// top plane of v_hi (context ctx_hi) to the upper one.  A rank that sweeps its interface cell layers first (into vectors of their
// own) starts the exchange with this before it sweeps its interior cells; stfem_halo_end then adds what arrived into the assembled
// destination vector as usual.
int stfem_halo_begin_split(stfem_comm *c, stfem_ctx *ctx_lo, stfem_vec *v_lo, stfem_ctx *ctx_hi, stfem_vec *v_hi, int lower, int upper, void *stream)
{
  if (!ctx_lo || !ctx_hi || !c || !v_lo || !v_hi || check_neighbours(c, lower, upper) != STFEM_OK || c->pending)
    return STFEM_ERR_INVALID_ARGUMENT;
  size_t pb, pb_hi;
  int nz_lo, nz;
  int rc = plane_geometry(ctx_lo, v_lo, pb, nz_lo);
  if (rc == STFEM_OK) rc = plane_geometry(ctx_hi, v_hi, pb_hi, nz);
  if (rc != STFEM_OK) return rc;
  if (pb != pb_hi || stfem_ctx_precision(ctx_lo) != stfem_ctx_precision(ctx_hi)) return STFEM_ERR_SHAPE_MISMATCH;
  stfem_ctx *ctx = ctx_hi;
  if (lower < 0 && upper < 0) {
    c->lower = lower; c->upper = upper; c->plane_bytes = pb;
    c->pending = true;
    return STFEM_OK;
  }
  COMM_HIP(hipSetDevice(c->device));
  if ((rc = ensure_buffers(c, pb)) != STFEM_OK) return rc;
  char *ts = static_cast<char *>(c->buf), *bs = ts + pb, *tr = bs + pb, *br = tr + pb;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // this rank's partial sums of its two interface planes
  if (upper >= 0 && (rc = stfem_plane_pack(ctx_hi, v_hi, nz - 1, ts, st)) != STFEM_OK) return rc;
  if (lower >= 0 && (rc = stfem_plane_pack(ctx_lo, v_lo, 0, bs, st)) != STFEM_OK) return rc;
  COMM_HIP(hipEventRecord(c->packed, st));
  COMM_HIP(hipStreamWaitEvent(c->stream, c->packed, 0));
  const ncclDataType_t dt = stfem_ctx_precision(ctx) ? ncclFloat : ncclDouble;
  const size_t count = pb / (stfem_ctx_precision(ctx) ? sizeof(float) : sizeof(double));
  COMM_NCCL(rccl().GroupStart());
  if (upper >= 0) {
    COMM_NCCL_IN_GROUP(rccl().Send(ts, count, dt, upper, c->comm, c->stream));
    COMM_NCCL_IN_GROUP(rccl().Recv(tr, count, dt, upper, c->comm, c->stream));
  }
  if (lower >= 0) {
    COMM_NCCL_IN_GROUP(rccl().Send(bs, count, dt, lower, c->comm, c->stream));
    COMM_NCCL_IN_GROUP(rccl().Recv(br, count, dt, lower, c->comm, c->stream));
  }
  COMM_NCCL(rccl().GroupEnd());
  COMM_HIP(hipEventRecord(c->arrived, c->stream));
  c->lower = lower; c->upper = upper; c->plane_bytes = pb; // (only a successful begin leaves an exchange pending)
  c->pending = true;
  return STFEM_OK;
}

int stfem_halo_end(stfem_ctx *ctx, stfem_comm *c, stfem_vec *v, void *stream)
{
  if (!ctx || !c || !v || !c->pending) return STFEM_ERR_INVALID_ARGUMENT;
  c->pending = false;
  if (c->lower < 0 && c->upper < 0) return STFEM_OK;
  size_t pb;
  int nz;
  int rc = plane_geometry(ctx, v, pb, nz);
  if (rc != STFEM_OK || pb != c->plane_bytes) return STFEM_ERR_SHAPE_MISMATCH;
  COMM_HIP(hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  COMM_HIP(hipStreamWaitEvent(st, c->arrived, 0));
  char *tr = static_cast<char *>(c->buf) + 2 * pb, *br = tr + pb;
  if (c->upper >= 0 && (rc = stfem_plane_unpack(ctx, v, nz - 1, tr, 1, st)) != STFEM_OK) return rc;
  if (c->lower >= 0 && (rc = stfem_plane_unpack(ctx, v, 0, br, 1, st)) != STFEM_OK) return rc;
  // the buffers are reused by the next exchange: it must not start before these adds have read them
  COMM_HIP(hipEventRecord(c->packed, st));
  COMM_HIP(hipStreamWaitEvent(c->stream, c->packed, 0));
  return STFEM_OK;
}

int stfem_ghost_update(stfem_ctx *ctx, stfem_comm *c, stfem_vec *v, int lower, int upper, void *stream)
{
  if (!ctx || !c || !v || check_neighbours(c, lower, upper) != STFEM_OK || c->pending)
    return STFEM_ERR_INVALID_ARGUMENT;
  if (lower < 0 && upper < 0) return STFEM_OK;
  size_t pb;
  int nz;
  int rc = plane_geometry(ctx, v, pb, nz);
  if (rc != STFEM_OK) return rc;
  COMM_HIP(hipSetDevice(c->device));
  if ((rc = ensure_buffers(c, pb)) != STFEM_OK) return rc;
  char *bs = static_cast<char *>(c->buf) + pb, *tr = bs + pb;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // the owner of an interface plane is the upper rank (its bottom plane); the lower rank's top plane is the ghost
  if (lower >= 0 && (rc = stfem_plane_pack(ctx, v, 0, bs, st)) != STFEM_OK) return rc;
  COMM_HIP(hipEventRecord(c->packed, st));
  COMM_HIP(hipStreamWaitEvent(c->stream, c->packed, 0));
  const ncclDataType_t dt = stfem_ctx_precision(ctx) ? ncclFloat : ncclDouble;
  const size_t count = pb / (stfem_ctx_precision(ctx) ? sizeof(float) : sizeof(double));
  COMM_NCCL(rccl().GroupStart());
  if (lower >= 0) COMM_NCCL_IN_GROUP(rccl().Send(bs, count, dt, lower, c->comm, c->stream));
  if (upper >= 0) COMM_NCCL_IN_GROUP(rccl().Recv(tr, count, dt, upper, c->comm, c->stream));
  COMM_NCCL(rccl().GroupEnd());
  COMM_HIP(hipEventRecord(c->arrived, c->stream));
  COMM_HIP(hipStreamWaitEvent(st, c->arrived, 0));
  if (upper >= 0 && (rc = stfem_plane_unpack(ctx, v, nz - 1, tr, 0, st)) != STFEM_OK) return rc;
  COMM_HIP(hipEventRecord(c->packed, st));
  COMM_HIP(hipStreamWaitEvent(c->stream, c->packed, 0));
  return STFEM_OK;
}

int stfem_dot_global(stfem_ctx *ctx, stfem_comm *c, const stfem_vec *a, const stfem_vec *b, int64_t n_own,
                     double *out, void *stream)
{
  if (!ctx || !c || !a || !b || !out) return STFEM_ERR_INVALID_ARGUMENT;
  double local = 0.0;
  const int rc = stfem_dot(ctx, a, b, n_own, &local, stream); // synchronous
  if (rc != STFEM_OK) return rc;
  if (c->world == 1) {
    *out = local;
    return STFEM_OK;
  }
  COMM_HIP(hipSetDevice(c->device));
  COMM_HIP(hipMemcpyAsync(c->d_red, &local, sizeof(double), hipMemcpyHostToDevice, c->stream));
  COMM_NCCL(rccl().AllReduce(c->d_red, c->d_red, 1, ncclDouble, ncclSum, c->comm, c->stream));
  COMM_HIP(hipMemcpyAsync(out, c->d_red, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  COMM_HIP(hipStreamSynchronize(c->stream));
  return STFEM_OK;
}

} // extern "C"
