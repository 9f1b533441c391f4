/*
 * stfem.h -- C-ABI of the MI355X-native matrix-free space-time operator apply.
 *
 * This is the drop-in boundary for ONE hot path of immaaane/dealii-stfem: the
 * space-time operator application  dst = (Alpha (x) K + Beta (x) M) src.
 * Every entry point names the reference interface it replaces (paths relative
 * to the reference repository).  The reference has no FFI: its boundary is the
 * duck-typed C++ operator concept (vmult/Tvmult/...) consumed by deal.II
 * solvers.  dealii-stfem_amd/host/stfem/operators.h holds header-only C++
 * classes with the reference's class and method names on top of this ABI;
 * INTEGRATION.md shows the deal.II-side glue.
 *
 * Conventions: plain pointers and sizes only; all functions return 0 on
 * success or a negative stfem_status; nothing throws across the boundary.
 * "device pointer" = HIP device memory of the context's device.  `stream` is a
 * hipStream_t passed as void* (NULL = the null stream).  Calls are
 * asynchronous on `stream` unless noted.  One host thread per context, and ONE
 * STREAM AT A TIME per context: a context owns device workspace (the sweeps'
 * halo slabs and tile counters, the transfers' temporaries) that two operations
 * of the same context running concurrently on different streams would share.
 * Work on different contexts may overlap freely.
 *
 * DoF numbering: lexicographic over the structured mesh,
 * index = ix + nx*(iy + ny*iz), nx = p*ncell[0]+1, x fastest.
 */
#ifndef STFEM_H
#define STFEM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  STFEM_OK = 0,
  STFEM_ERR_INVALID_ARGUMENT = -1,
  STFEM_ERR_UNSUPPORTED = -2, /* degree / block count / mode not built */
  STFEM_ERR_HIP = -3,         /* a HIP runtime call failed: see stfem_last_hip_error */
  STFEM_ERR_NO_DEVICE = -4,
  STFEM_ERR_SHAPE_MISMATCH = -5,
  STFEM_ERR_ALIAS = -6, /* vmult(dst, src) with dst == src (reference forbids it too) */
  STFEM_ERR_OUT_OF_MEMORY = -7,
  STFEM_ERR_COMM = -8 /* an RCCL call failed: see stfem_comm_last_error */
} stfem_status;

typedef struct stfem_ctx stfem_ctx; /* replaces MatrixFree + MatrixFreeOperator state */
typedef struct stfem_vec stfem_vec; /* replaces LinearAlgebra::distributed::BlockVector */

/* Mesh of one rank: a structured block of hexahedra (MappingQ1 geometry).
 * Replaces GridGenerator::subdivided_hyper_rectangle + refine_global +
 * GridTools::distort_random as used by tests/tp_01.cc:82-90. */
typedef struct {
  int32_t ncell[3];       /* cells per direction on this rank */
  const double *vertices; /* host, (ncell+1)^3 * 3, x fastest, xyz interleaved; NULL = Cartesian box */
  double lower[3];        /* used when vertices == NULL */
  double upper[3];
  int32_t dirichlet_mask; /* bit0 -x, bit1 +x, bit2 -y, bit3 +y, bit4 -z, bit5 +z: faces carrying
                             homogeneous Dirichlet constraints (make_zero_boundary_constraints,
                             tests/tp_01.cc:98).  Partition interfaces are NOT set. */
  int32_t device;         /* HIP device ordinal */
} stfem_mesh_desc;

/* Replaces FE_Q<dim>(p) + QGauss<dim>(p+1) of tests/tp_01.cc:76-77. */
typedef struct {
  int32_t degree;       /* p, 1..5 (FE_Q(5): the tile sweep on every mesh; cell-patch smoother for one or two temporal blocks,
                         * i.e. cell blocks of up to 512 rows as for every degree) */
  int32_t n_q_points_1d; /* must be degree+1 */
  int32_t n_components; /* must be 1 (scalar path) */
  int32_t precision;    /* 0 = fp64 (the solver's Number, operators.cc:5-45), 1 = fp32 (the multigrid
                           levels' NumberPreconditioner, tests/tp_01.cc:780, 801-806): element type of
                           all device vectors, coefficients and metric terms of this context */
} stfem_space_desc;

/* MatrixFreeOperator ctor (include/operators.h:973-1004): builds device tables. Synchronous. */
int stfem_ctx_create(const stfem_mesh_desc *mesh, const stfem_space_desc *space, stfem_ctx **out);
void stfem_ctx_destroy(stfem_ctx *ctx);
/* MatrixFreeOperator::m() (operators.h:1047-1051): spatial DoFs on this rank */
int64_t stfem_n_dofs(const stfem_ctx *ctx);
int64_t stfem_n_cells(const stfem_ctx *ctx);
/* DoFs per direction of this rank's box, nd[d] = degree * ncell[d] + 1 (x fastest in every block) */
int stfem_n_dofs_1d(const stfem_ctx *ctx, int32_t nd[3]);
/* 1 if the mesh was recognised as an axis-aligned uniform box (Cartesian fast path) */
int stfem_is_cartesian(const stfem_ctx *ctx);
/* 0 = fp64, 1 = fp32: the Number type of the operator (MatrixFreeOperator<dim, n_components, Number>) */
int stfem_ctx_precision(const stfem_ctx *ctx);

/* MatrixFreeOperator::evaluate_coefficient (operators.h:1060-1087).
 * which: 0 = mass coefficient (the M operator), 1 = laplace coefficient (the K operator).
 * layout: 0 = clear (scalings 1 are used), 1 = one value per cell, 2 = one value per
 * (cell, quadrature point), host array [cell*nq^3 + q], q = qx + nq*(qy + nq*qz).
 * As in the reference the coefficient REPLACES the scaling.  Synchronous. */
int stfem_set_coefficient(stfem_ctx *ctx, int which, int layout, const double *host_values);

/* initialize_dof_vector (operators.h:1006-1011, 648-662): n_blocks spatial vectors of
 * stfem_n_dofs elements of the context's Number type each, zero-initialised. */
int stfem_vector_create(stfem_ctx *ctx, int n_blocks, stfem_vec **out);
/* View caller-owned device memory (e.g. the arrays behind a deal.II
 * LinearAlgebra::distributed::BlockVector<double, MemorySpace::Default>): one device pointer
 * per block, each stfem_n_dofs doubles.  Not freed by stfem_vector_destroy. */
int stfem_vector_wrap(stfem_ctx *ctx, int n_blocks, void *const *device_blocks, stfem_vec **out);
/* Point an existing view at other device arrays (the hot path of a binding keeps one view per argument
 * and rebinds it per vmult: no allocation).  Only views made by stfem_vector_wrap can be rebound. */
int stfem_vector_rebind(stfem_vec *v, int n_blocks, void *const *device_blocks);
void stfem_vector_destroy(stfem_vec *v);
int stfem_vector_n_blocks(const stfem_vec *v);
void *stfem_vector_block(const stfem_vec *v, int block); /* device pointer */
/* host <-> device copies of all blocks (host_blocks[b] has stfem_n_dofs doubles; fp32 contexts
 * convert on the way). Synchronous. */
int stfem_vector_upload(stfem_vec *v, const double *const *host_blocks);
int stfem_vector_download(const stfem_vec *v, double *const *host_blocks);

/* SystemMatrix::vmult / Tvmult / vmult_slice / vmult_slice_add
 * (include/operators.h:536-559, 561-583, 377-382, 586-611), fused into one cell sweep:
 *   transpose == 0:  dst_j (+)= sum_i alpha[j*ncols+i] K src_i + beta[j*ncols+i] M src_i,
 *                    j < nrows, i < ncols      (src has ncols blocks, dst has nrows)
 *   transpose != 0:  dst_j (+)= sum_i alpha[i*ncols+j] K src_i + beta[i*ncols+j] M src_i,
 *                    i < nrows, j < ncols      (src has nrows blocks, dst has ncols)
 * alpha, beta: host, row-major nrows x ncols (FullMatrix<Number> Alpha, Beta).
 * add == 0 overwrites dst (the reference's `dst = 0.0`), add != 0 accumulates
 * (vmult_slice_add).  ncols == 1 is the reference's n x 1 "slice" case.
 * K = MatrixFreeOperator(0,1), M = MatrixFreeOperator(1,0) with the coefficients set on ctx.
 * Constrained rows of dst are left untouched (add) or zero (overwrite). */
int stfem_st_vmult(stfem_ctx *ctx, int nrows, int ncols, const double *alpha, const double *beta,
                   int transpose, int add, stfem_vec *dst, const stfem_vec *src, void *stream);

/* MatrixFreeOperator::vmult (operators.h:1013-1018): dst = (ms*M_c + ls*K_c) src, one block. */
int stfem_space_vmult(stfem_ctx *ctx, double mass_scaling, double laplace_scaling, stfem_vec *dst,
                      const stfem_vec *src, void *stream);

/* compute_diagonal / get_matrix_diagonal (operators.h:1092-1110, 1035-1039), forward diagonal
 * of ms*M_c + ls*K_c into block 0 of diag; constrained rows are 0. */
int stfem_diagonal(stfem_ctx *ctx, double mass_scaling, double laplace_scaling, stfem_vec *diag,
                   void *stream);
/* get_matrix_diagonal_inverse (operators.h:1106-1109): 1 / d where |d| > sqrt(epsilon of the
 * operator's Number), 1 elsewhere (constrained rows). */
int stfem_diagonal_inverse(stfem_ctx *ctx, double mass_scaling, double laplace_scaling,
                           stfem_vec *diag, void *stream);
/* SystemMatrix::get_matrix_diagonal (operators.h:613-623): block i of diag =
 * Alpha(i,i) * diag K + Beta(i,i) * diag M with K = MatrixFreeOperator(0,1), M = (1,0);
 * inverse != 0: get_matrix_diagonal_inverse exactly as the reference combines it (633-637):
 * 1/Alpha(i,i) * (diag K)^-1 + 1/Beta(i,i) * (diag M)^-1.  alpha, beta: n x n row-major. */
int stfem_st_diagonal(stfem_ctx *ctx, int n, const double *alpha, const double *beta, int inverse,
                      stfem_vec *diag, void *stream);

/* Block BLAS-1 used around the operator (operators.h:211-283 tensorproduct_add;
 * LinearAlgebra::distributed::Vector::add / l2_norm / operator*).  Local to this rank:
 * the caller all-reduces `out` over ranks (RCCL/MPI).  dot/norm are synchronous.
 * n_own limits the reduction to the first n_own entries of every block (owned range). */
int stfem_tensorproduct_add(stfem_ctx *ctx, int nrows, int ncols, const double *A, stfem_vec *c,
                            const stfem_vec *b, void *stream);
int stfem_dot(stfem_ctx *ctx, const stfem_vec *a, const stfem_vec *b, int64_t n_own, double *out,
              void *stream);
/* The Gram-Schmidt step of the Krylov solvers (SolverFGMRES / SolverGMRES of deal.II: k inner products and k vector updates per
 * iteration) in a few launches and ONE read-back.  All reductions are two-stage with a fixed summation order: bitwise reproducible.
 *   multi_dot:     out[i] = <a_i, b>, i < k (host array; synchronous)
 *   multi_axpy:    y += sum_i coef_i x_i (coefficients from the host; asynchronous)
 *   orthogonalize: one classical Gram-Schmidt pass h = V^T w, w -= V h with the coefficients kept on the device, then
 *                  h_out[0 .. k) <- h; if given, norm2_before <- <w, w> before the projection (rides in the launch of the
 *                  coefficients) and norm2_out <- <w, w> after it (synchronous).  Twice in a row it is the re-orthogonalised
 *                  classical scheme (as stable as the modified one, k times fewer launches and waits); the second pass is only
 *                  needed when the projection cancelled most of w (norm2_out << norm2_before), which the two norms tell. */
int stfem_multi_dot(stfem_ctx *ctx, int k, const stfem_vec *const *a, const stfem_vec *b, int64_t n_own, double *out, void *stream);
int stfem_multi_axpy(stfem_ctx *ctx, int k, const double *coef, const stfem_vec *const *x, stfem_vec *y, void *stream);
int stfem_orthogonalize(stfem_ctx *ctx, int k, const stfem_vec *const *v, stfem_vec *w, int64_t n_own, double *h_out, double *norm2_before,
                        double *norm2_out, void *stream);

/* Halo support for z-slab partitions (deal.II: update_ghost_values / compress(add) inside
 * MatrixFree::cell_loop, operators.h:1016-1017).  A rank's top DoF plane (iz = nz-1) is the
 * ghost copy of its upper neighbour's bottom plane (iz = 0).
 * pack:   buf[b*plane + i] = v_b[plane_index(iz) + i]        (plane = nx*ny doubles)
 * unpack: v_b[plane_index(iz) + i]  = or +=  buf[b*plane + i]
 * buf is a device pointer with n_blocks*nx*ny doubles. */
int stfem_plane_pack(stfem_ctx *ctx, const stfem_vec *v, int iz, void *device_buf, void *stream);
int stfem_plane_unpack(stfem_ctx *ctx, stfem_vec *v, int iz, const void *device_buf, int add,
                       void *stream);

/* The exchange itself, over RCCL, for callers that do not bring their own transport: one process per
 * GPU, ranks = z-slabs of the mesh.  Replaces the MPI side of LinearAlgebra::distributed::Vector
 * (update_ghost_values / compress(add), operators.h:1016-1017) and of Vector::operator* (MPI_Allreduce).
 * The 128-byte id is created on one rank and broadcast by the host code (MPI_Bcast on the deal.II side),
 * then every rank creates its communicator (collective).  RCCL is bound at run time: without librccl.so
 * stfem_comm_create returns STFEM_ERR_UNSUPPORTED.  lower_rank / upper_rank: the ranks owning the slab
 * below / above, -1 at the ends of the mesh. */
#define STFEM_COMM_ID_BYTES 128
typedef struct stfem_comm stfem_comm;
int stfem_comm_get_unique_id(void *id);
int stfem_comm_create(const void *id, int rank, int world, int device, stfem_comm **out);
void stfem_comm_destroy(stfem_comm *comm);
/* 1 if an RCCL could be bound in this process (ask on EVERY rank and agree before any rank enters the collective
 * stfem_comm_create); the rank count RCCL itself reports for the communicator (ncclCommCount; 0 if unknown) */
int stfem_comm_available(void);
int stfem_comm_rccl_count(const stfem_comm *comm);
int stfem_comm_rank(const stfem_comm *comm);
int stfem_comm_size(const stfem_comm *comm);
const char *stfem_comm_last_error(void);
/* src.update_ghost_values(): the top plane of every block <- the upper neighbour's bottom plane.
 * Enqueued on `stream` (the transfer itself runs on the communicator's own stream in between). */
int stfem_ghost_update(stfem_ctx *ctx, stfem_comm *comm, stfem_vec *v, int lower_rank, int upper_rank,
                       void *stream);
/* dst.compress(add) + update_ghost_values in one exchange: after the local cell sweep both copies of an
 * interface plane hold partial sums; each side sends its partial to the other and adds what it receives,
 * all temporal blocks in one message per neighbour.  begin: packs on `stream` and starts the transfers on
 * the communicator's stream; work enqueued on `stream` between begin and end overlaps with them;
 * end: `stream` waits for the arrival and adds.  One exchange in flight per communicator. */
int stfem_halo_begin(stfem_ctx *ctx, stfem_comm *comm, stfem_vec *v, int lower_rank, int upper_rank,
                     void *stream);
int stfem_halo_end(stfem_ctx *ctx, stfem_comm *comm, stfem_vec *v, void *stream);
/* The exchange with the two interface planes taken from two vectors (plane 0 of v_lo goes down, the top plane of v_hi goes up): for
 * a rank that sweeps its interface cell layers first, into vectors of their own, and its interior cells while the planes travel -
 * deal.II's cell_loop overlaps the ghost exchange with the cells that do not need it (include/operators.h:1016-1017).
 * stfem_halo_end on the assembled destination vector completes it.  stfem_planes_move copies nplanes consecutive DoF planes between
 * vectors of two contexts with the same plane size (add_mask bit 0 / 1: the first / last plane is added instead): how the interface
 * layers' results reach the destination vector (dealii-stfem_amd/distributed.py: OverlappedSlabOperator). */
int stfem_halo_begin_split(stfem_comm *comm, stfem_ctx *ctx_lo, stfem_vec *v_lo, stfem_ctx *ctx_hi, stfem_vec *v_hi, int lower, int upper,
                           void *stream);
int stfem_planes_move(stfem_ctx *ctx_src, const stfem_vec *src, int iz_src, stfem_ctx *ctx_dst, stfem_vec *dst, int iz_dst, int nplanes,
                      int add_mask, void *stream);
/* stfem_dot followed by the sum over all ranks (synchronous) */
int stfem_dot_global(stfem_ctx *ctx, stfem_comm *comm, const stfem_vec *a, const stfem_vec *b,
                     int64_t n_own, double *out, void *stream);

/* Time-multigrid transfer matrices (include/fe_time.h:749-898: get_time_prolongation_matrix,
 * get_time_restriction_matrix, get_time_projection_matrix), row-major dims[0] x dims[1]; `out` may be NULL to ask
 * for the dimensions only.  type: 0 = cG, 1 = dG.  Prolongation / restriction couple n_timesteps_at_once fine steps
 * with half as many of twice the length (a power of two >= 2); the projection changes the temporal degree.  They act
 * on block vectors through stfem_tensorproduct_add (the reference: tensorproduct_add in MGTwoLevelTransferTime). */
int stfem_time_prolongation_matrix(int type, int r, int n_timesteps_at_once, double *out, int32_t dims[2]);
int stfem_time_restriction_matrix(int type, int r, int n_timesteps_at_once, double *out, int32_t dims[2]);
int stfem_time_projection_matrix(int type, int r_src, int r_dst, int n_timesteps_at_once, double *out, int32_t dims[2]);

/* Level schedule of the space-time multigrid (include/fe_time.cc:17-150).  Transfer kinds are the characters of the
 * reference's MGType: 't' (tau: half the time steps per slab), 'k' (temporal degree), 'h' (mesh), 'p' (spatial degree);
 * sequences list the coarsest transfer first.
 * poly_mg_sequence: get_poly_mg_sequence(k_max, k_min, type), type 0 = bisect, 1 = decrease_by_one, 2 = go_to_one.
 * mg_sequence: get_mg_sequence(n_sp_lvl, k_seq, p_seq, ...) - only the lengths n_k, n_p of the degree sequences
 *   enter; lower_lvl 'k' or 't'; coarsening_type 0 = space_or_time, 1 = space_and_time (CoarseningType, types.h:102).
 * precondition_stmg_types: get_precondition_stmg_types - n + 1 smoother ids (0 = identity: the level does not smooth).
 * `out` may be NULL to ask for the length only (first two). */
int stfem_poly_mg_sequence(int k_max, int k_min, int sequence_type, int32_t *out, int32_t *n_out);
int stfem_mg_sequence(int n_sp_lvl, int n_k, int n_p, int n_timesteps_at_once, int n_timesteps_at_once_min, char lower_lvl,
                      int coarsening_type, int time_before_space, int use_p_multigrid_space, int zip_from_back, char *out,
                      int32_t *n_out);
int stfem_precondition_stmg_types(const char *mg_type_level, int n, int coarsening_type, int time_before_space, int smoother,
                                  int32_t *out);

/* Space transfer between two levels (deal.II MGTwoLevelTransfer as the reference uses it: include/stmg.h:38-110,
 * built in build_stmg_transfers, stmg.h:580-600): `fine` has per direction the same or twice the cells of `coarse`
 * and a degree >= the coarse one (h-, p- or hp-transfer); both contexts on one device, same precision, their
 * dirichlet masks are the constraints of the two levels.  All blocks of a block vector are transferred alike
 * (MGTwoLevelBlockTransfer).
 *   prolongate:  dst_fine (+)= P src_coarse  (embedding of the coarse space; constrained fine rows stay 0 / untouched)
 *   restrict:    dst_coarse (+)= P^T src_fine  (restrict_and_add is add = 1)
 *   interpolate: dst_coarse = the fine function at the coarse nodes (MGTwoLevelTransfer::interpolate)
 * Asynchronous on `stream`; one transfer object serves one stream at a time (it owns the intermediates). */
typedef struct stfem_transfer stfem_transfer;
int stfem_transfer_create(stfem_ctx *fine, stfem_ctx *coarse, stfem_transfer **out);
/* the same between the two levels of one z-slab of a partitioned mesh (both contexts hold the slab: the coarse one half the cell
 * layers or the same); neighbour_mask: bit 4 = a slab below, bit 5 = a slab above.  The prolongation is local (the coarse ghost
 * plane must be up to date: stfem_ghost_update).  The restriction leaves PARTIAL sums in the coarse interface planes, to be
 * completed by the add-exchange (stfem_halo_begin / end on the coarse context); the fine ghost plane (top plane under a slab
 * above) is restricted by its owner only. */
int stfem_transfer_create_partitioned(stfem_ctx *fine, stfem_ctx *coarse, int neighbour_mask, stfem_transfer **out);
void stfem_transfer_destroy(stfem_transfer *t);
int stfem_transfer_prolongate(stfem_transfer *t, stfem_vec *dst_fine, const stfem_vec *src_coarse, int add, void *stream);
int stfem_transfer_restrict(stfem_transfer *t, stfem_vec *dst_coarse, const stfem_vec *src_fine, int add, void *stream);
int stfem_transfer_interpolate(stfem_transfer *t, stfem_vec *dst_coarse, const stfem_vec *src_fine, void *stream);
const char *stfem_transfer_last_error(void);
/* the 1D factors a transfer is the Kronecker product of, without constraints (host only, for checks):
 * P [n_f x n_c] embedding, I [n_c x n_f] nodal interpolation, n = degree * ncell + 1; either may be NULL */
int stfem_transfer_line_matrices(int ncell_fine, int degree_fine, int ncell_coarse, int degree_coarse, double *P, double *I);
/* dst = src between vectors of two contexts with the same number of DoFs, converting between fp64 and fp32
 * (GMG::vmult, stmg.h:1330-1343: the multigrid runs in NumberPreconditioner, the solver in Number) */
int stfem_vector_convert(stfem_vec *dst, const stfem_vec *src, void *stream);

/* A launch-bound fixed sequence of calls (one multigrid V-cycle is ~1300 kernel launches, most of them on coarse levels)
 * recorded once and replayed as one hipGraph.  stream_create: a stream of the caller's own (ordered against the default
 * stream the other entry points use when given NULL).  graph_begin(stream): from here every stfem_* call given this stream
 * is recorded, not run (calls that allocate or synchronise - vector_create, dot, upload, the first use of an operator on a
 * context - are not allowed in between: run the sequence once before recording it); graph_end returns the instantiated
 * graph; graph_launch replays it on `stream` with the vectors (device pointers) it was recorded with.  Errors:
 * stfem_transfer_last_error. */
typedef struct stfem_graph stfem_graph;
int stfem_stream_create(void **stream_out);
void stfem_stream_destroy(void *stream);
int stfem_stream_synchronize(void *stream);
int stfem_graph_begin(void *stream);
int stfem_graph_end(void *stream, stfem_graph **out);
int stfem_graph_launch(stfem_graph *g, void *stream);
void stfem_graph_destroy(stfem_graph *g);

/* Cell-patch Vanka / additive-Schwarz smoother of the space-time system A = Alpha (x) K + Beta (x) M:
 * PreconditionVanka (include/stmg.h:619-907; set-up 786-829 with compute_block_matrix.h:50-139, apply
 * 832-872).  create: builds and inverts the valence-weighted cell blocks of the ASSEMBLED matrices (zero
 * boundary constraints as in tests/tp_01.cc:283-299); Alpha, Beta are n x n row-major.  One rank.  Axis-aligned
 * uniform meshes without coefficient tables hold one block per neighbour pattern (at most 27); every other
 * context one block per cell (host-side set-up: STFEM_ERR_OUT_OF_MEMORY beyond 64 GB of blocks).  vmult: dst = sum over cells of scatter(B_c^-1 gather(src)),
 * dst is overwritten, dst must not alias src (as in the reference). */
typedef struct stfem_vanka stfem_vanka;
int stfem_vanka_create(stfem_ctx *ctx, int n, const double *alpha, const double *beta, stfem_vanka **out);
/* the same on one slab of a partitioned mesh: neighbour_mask (bits as dirichlet_mask) names the faces behind which another rank
 * holds the next cells.  The blocks and valences of the cells at such a face count the cells beyond it, as the reference's do on
 * a parallel::distributed::Triangulation (ghost cells); stfem_vanka_vmult then leaves PARTIAL sums in the interface planes of
 * dst, to be completed by the add-exchange of the operator (stfem_halo_begin / end).  Axis-aligned uniform meshes; for
 * one-block-per-cell contexts (STFEM_ERR_UNSUPPORTED here) see stfem_vanka_create_partitioned_general. */
int stfem_vanka_create_partitioned(stfem_ctx *ctx, int n, const double *alpha, const double *beta, int neighbour_mask, stfem_vanka **out);
/* The same on a z-slab of a GENERAL mesh (perturbed cells, coefficient tables - BASELINE configs[2] on several ranks), where the
 * blocks of the cells next to an interface need the cell matrices of the neighbour rank's cells (the reference builds its blocks
 * on owned and ghost cells, stmg.h:688-689, 795-796; compute_block_matrix.h:68-73).  `extended` is a context of the same slab
 * plus ONE ghost cell layer on every z side named in neighbour_mask (bit 4: lower, bit 5: upper) - same degree, precision and
 * x-y constraints, no constraint on a z face with a ghost layer; the vertices of the ghost layers are all that crosses the
 * ranks.  It may be destroyed after the call.  On axis-aligned uniform meshes this is stfem_vanka_create_partitioned. */
int stfem_vanka_create_partitioned_general(stfem_ctx *slab, stfem_ctx *extended, int n, const double *alpha, const double *beta,
                                           int neighbour_mask, stfem_vanka **out);
void stfem_vanka_destroy(stfem_vanka *v);
int stfem_vanka_n_classes(const stfem_vanka *v); /* distinct cell blocks held */
/* diagnostics: {row tiles (16 rows) per workgroup, parts per cell block} */
int stfem_vanka_plan(const stfem_vanka *v, int32_t out[2]);
int stfem_vanka_vmult(stfem_vanka *v, stfem_vec *dst, const stfem_vec *src, void *stream);
/* dst = (accumulate ? dst : 0) + omega * V src: the step x <- x + omega P^-1 r of PreconditionRelaxation around the smoother
 * (include/stmg.h:1199-1238) fused into the smoother's scatter - no temporary vector, no separate update pass. */
int stfem_vanka_step(stfem_vanka *v, stfem_vec *dst, double omega, int accumulate, const stfem_vec *src, void *stream);
const char *stfem_vanka_last_error(void);

/* Around the operator, for the slab driver (include/time_integrators.h, tests/tp_01.cc:382-400, 646-725,
 * include/exact_solution.h:503-649).  The caller evaluates its functions at the points and hands the values in
 * (host arrays); nq = points of the Gauss rule per direction (tests/tp_01.cc uses degree + 1 for the load vector,
 * ErrorCalculator one fewer).
 * support_points:    out[ndofs][3], the nodes in the order of the vectors (VectorTools::interpolate = evaluate + upload)
 * quadrature_points: out[cell][q][3], q = qx + nq (qy + nq qz)
 * integrate_rhs:     block `block` of dst = (f, phi_i) with QGauss(nq), constrained rows 0
 *                    (VectorTools::create_right_hand_side with the zero-boundary constraints).  Synchronous.
 * integrate_difference: out = { sum JxW (u_h - u)^2, max |u_h - u|, sum JxW |grad u_h - grad u|^2 } over the
 *                    quadrature points (VectorTools::integrate_difference for L2_norm squared, Linfty_norm,
 *                    H1_seminorm squared); exact_grad_at_points [cell][q][3] may be NULL (third entry 0).  Synchronous.
 * vector_axpby:      y = a x + b y on every block.  A zero factor means "not read": b = 0 overwrites y (equ), a = 0 never
 *                    touches x (a = b = 0 assigns zero whatever y held, NaN included); x and y may be the same vector
 * vector_set_zero:   y = 0 (`dst = 0.0` of the reference's smoother, include/stmg.h:837) */
int stfem_support_points(const stfem_ctx *ctx, double *out);
int stfem_quadrature_points(const stfem_ctx *ctx, int nq, double *out);
int stfem_integrate_rhs(stfem_ctx *ctx, int nq, const double *f_at_points, stfem_vec *dst, int block, void *stream);
int stfem_integrate_difference(stfem_ctx *ctx, int nq, const stfem_vec *u, int block, const double *exact_at_points,
                               const double *exact_grad_at_points, double out[3], void *stream);
/* the same two with f(x) = amplitude * prod_d sin(2 pi frequency x_d) - the separable right-hand sides and exact solutions of the reference's
 * convergence tests (include/exact_solution.h:27-81, 147-197) at a fixed time - evaluated on the device: nothing crosses the host */
int stfem_integrate_rhs_product(stfem_ctx *ctx, int nq, double amplitude, double frequency, stfem_vec *dst, int block, void *stream);
int stfem_integrate_difference_product(stfem_ctx *ctx, int nq, const stfem_vec *u, int block, double amplitude, double frequency, double out[3],
                                       void *stream);
int stfem_vector_axpby(stfem_ctx *ctx, double a, const stfem_vec *x, double b, stfem_vec *y, void *stream);
int stfem_vector_set_zero(stfem_ctx *ctx, stfem_vec *y, void *stream);
/* the same arithmetic on n_arrays device arrays of len[i] elements of the context's Number in ONE launch: y_i = a x_i + b y_i with the
 * zero-factor rules of stfem_vector_axpby (x may be NULL when a = 0).  For vectors whose blocks differ in length - the velocity and
 * pressure blocks of the Stokes system (BlockVectorT::sadd / equ / = 0 over all blocks of a two-variable vector) */
int stfem_axpby_many(stfem_ctx *ctx, int n_arrays, const int64_t *len, double a, const void *const *x, double b, void *const *y, void *stream);
const char *stfem_driver_last_error(void);
/* QGauss(n) on [0, 1]; the support points of the temporal basis, get_time_quad (fe_time.cc:152-161): QGaussLobatto(r + 1)
 * for cG(r) (type 0), QGaussRadau(r + 1, right) for dG(r) (type 1); r + 1 values */
int stfem_gauss_rule(int n, double *points, double *weights);
int stfem_fe_time_points(int type, int r, double *points);

/* Host-side helpers mirroring include/fe_time.h (type: 0 = CGP, 1 = DG).  Row-major outputs,
 * nb = (type==0 ? r : r+1) * n_timesteps_at_once; returns nb or a negative status.
 * get_fe_time_weights (fe_time.h:351-409) */
int stfem_fe_time_weights(int type, int r, double time_step_size, int n_timesteps_at_once,
                          double *Alpha, double *Beta, double *Gamma, double *Zeta);
/* get_fe_time_weights_wave (fe_time.h:157-305) on the single-step matrices of the above */
int stfem_fe_time_weights_wave(int type, int r, double time_step_size, int n_timesteps_at_once,
                               double *Alpha_lhs, double *Beta_lhs, double *rhs_uK,
                               double *rhs_uM, double *rhs_vM);
/* Structured vertex grid with optional interior-vertex jitter (stand-in for
 * GridTools::distort_random, tests/tp_01.cc:89-90; own mt19937_64 stream):
 * out has (ncell+1)^3*3 doubles.  offset_cells/global_ncell select a z-slab of a global mesh
 * so that all ranks see one consistent perturbation. */
int stfem_mesh_vertices(const int32_t global_ncell[3], const double lower[3],
                        const double upper[3], double distort, uint64_t seed, int32_t z_cell_begin,
                        int32_t z_cell_end, double *out);
/* Coefficient<dim>::value at cell centres (operators.h:870-965): one value per cell. */
int stfem_coefficient_per_cell(const int32_t ncell[3], const double *vertices, double c1, double c2,
                               double c3, double distort_coeff, const int32_t subdivisions[3],
                               const double lower[3], const double upper[3], double *out);

/* ---- Stokes two-field operator (BASELINE configs[4]): the cell loop (LoopType::Cell, include/operators.h:1228-1229) and,
 * after stfem_stokes_set_weak_boundaries below, the weak (Nitsche) boundary faces of operators.h:1662-1751; the CIP interior-face
 * term (delta0 != 0, 1603-1638) and the convection modes are not built.  Velocity FE_Q(2)^3, pressure FE_Q(1) (stfem_stokes_create)
 * or FE_DGP(1) (stfem_stokes_create_ex), QGauss(3), MappingQ1 on the mesh of `mesh`; mesh->dirichlet_mask constrains the velocity
 * (homogeneous), the pressure is unconstrained.  fp64.
 * Layout: a velocity vector is 3 * n_velocity_dofs doubles, component-major, every component in
 * the scalar FE_Q(2) numbering of this header; a pressure vector is n_pressure_dofs doubles in
 * the scalar FE_Q(1) numbering.  All vector arguments are DEVICE pointers. */
typedef struct stfem_stokes_ctx stfem_stokes_ctx; /* replaces MatrixFree + StokesMatrixFreeOperator state */
/* StokesMatrixFreeOperator ctor (operators.h:1200-1251); only velocity_degree == 2 is built */
int stfem_stokes_create(const stfem_mesh_desc *mesh, int velocity_degree, double viscosity,
                        stfem_stokes_ctx **out);
/* the same with the pressure space chosen as tests/tp_03stokes.cc:83-86 does with dGPressure: 0 = FE_Q(degree - 1) (continuous,
 * BASELINE configs[4]), 1 = FE_DGP(degree - 1), the reference's default in tests/json/stokes.json: discontinuous, deal.II's basis
 * of orthonormal Legendre polynomials on the reference cell (degree 1: 1, l(xi), l(eta), l(zeta), l(x) = sqrt 3 (2 x - 1)), four
 * DoFs per cell numbered cell by cell (p[4 cell + j], cells lexicographic) */
int stfem_stokes_create_ex(const stfem_mesh_desc *mesh, int velocity_degree, int pressure_space, double viscosity, stfem_stokes_ctx **out);
void stfem_stokes_destroy(stfem_stokes_ctx *ctx);
int64_t stfem_stokes_n_velocity_dofs(const stfem_stokes_ctx *ctx); /* per component */
int64_t stfem_stokes_n_pressure_dofs(const stfem_stokes_ctx *ctx);
/* StokesMatrixFreeOperator::initialize_dof_vector(vec, variable) (operators.h:1254-1275):
 * variable 0 = velocity (3 * n_velocity_dofs doubles), 1 = pressure; zero-initialised device
 * memory.  upload / download copy a whole vector from / to the host and are synchronous. */
int stfem_stokes_vector_create(stfem_stokes_ctx *ctx, int variable, double **device_out);
void stfem_stokes_vector_destroy(stfem_stokes_ctx *ctx, double *device_vec);
int stfem_stokes_vector_upload(stfem_stokes_ctx *ctx, int variable, double *device_vec, const double *host);
int stfem_stokes_vector_download(stfem_stokes_ctx *ctx, int variable, const double *device_vec, double *host);
/* StokesMatrixFreeOperator::vmult (operators.h:1501-1575, OperatorMode::none):
 *   dst_u = nu K u - B^T p,   dst_p = B u      (B u = (div u, q)) */
int stfem_stokes_vmult(stfem_stokes_ctx *ctx, double *dst_u, double *dst_p, const double *src_u,
                       const double *src_p, void *stream);
/* the MassMatrixType of SystemMatrixStokes (vector mass, operators.h:1013-1018 with
 * n_components = dim): dst_u = M u */
int stfem_stokes_mass_vmult(stfem_stokes_ctx *ctx, double *dst_u, const double *src_u, void *stream);
/* SystemMatrixStokes::vmult (operators.h:696-700, 825-867): blocks are numbered by
 * BlockSlice::index(timestep, variable, timedof) (fe_time.h:956-967; variable 0 = velocity,
 * 1 = pressure), Alpha/Beta are the host row-major (2*nt*ns)^2 matrices of
 * get_fe_time_weights_stokes (fe_time.h:1242-1285).  For every source time dof (it, id):
 *   dst[index(jt,v,jd)] += Alpha(index(jt,v,jd), index(it,0,id)) * (K_S (u,p))_v
 *   dst[index(jt,0,jd)] += Beta (index(jt,0,jd), index(it,0,id)) * M u
 * entries with |.| <= 10 eps are skipped as in internal::scatter (operators.h:91-110); dst is
 * zeroed first.  Up to four time dofs: one set of colour launches for the whole system (every cell is evaluated for all sources, the
 * weighted results are summed in registers, every destination is written once); more: one fused launch per source time dof.  The reference's Tvmult for this class
 * (operators.h:702-745) indexes dst by the source time dof and is not a transpose: it is this entry with the effective matrices of
 * that rule (SystemMatrixStokes::Tvmult in host/stfem/stokes.h builds them). */
int stfem_stokes_st_vmult(stfem_stokes_ctx *ctx, int n_timesteps_at_once, int n_timedofs,
                          int variable_major, const double *Alpha, const double *Beta,
                          double *const *dst_blocks, const double *const *src_blocks, void *stream);
/* SystemMatrixStokes::vmult_slice_add (operators.h:748-781), the n x 1 case of the right-hand side:
 * Gamma, Zeta are host vectors of 2*nt*ns entries (column 0 of the reference's n x 1 matrices);
 *   dst[index(it,v,id)] += Gamma[index(it,v,id)] * (K_S (u,p))_v ,  dst[index(it,0,id)] += Zeta[index(it,0,id)] * M u
 * dst is not zeroed. */
int stfem_stokes_st_vmult_slice_add(stfem_stokes_ctx *ctx, int n_timesteps_at_once, int n_timedofs,
                                    int variable_major, const double *Gamma, const double *Zeta,
                                    double *const *dst_blocks, const double *src_u,
                                    const double *src_p, void *stream);
/* Weak boundary conditions of StokesMatrixFreeOperator (reference include/operators.h:1206-1211, 1220-1221, 1640-1741) and
 * StokesNitscheMatrixFreeOperator (1768-1951), linear operator (NonlinearTreatment::None).
 * Faces are numbered f = 2 d + s (direction d, side s: 0 = lower, 1 = upper; bit f of the masks) - the boundary ids of deal.II's
 * colorized hyper_rectangle, the same bits as stfem_mesh_desc.dirichlet_mask (strong constraints; keep the sets disjoint).
 *   set_weak_boundaries: faces in weak_mask get the Nitsche terms (gamma1 = viscosity penalty1, gamma2 = penalty2, h = sqrt(face
 *       area)) in every later stfem_stokes_vmult / st_vmult / st_vmult_slice_add (LoopType::Full); faces in outflow_mask add
 *       nothing to the linear operator (the reference's back-flow term carries a factor 0.0, the rest is nonlinear-only) and
 *       take precedence over weak_mask, as the reference checks them first (operators.h:1680).
 *   n_face_points / face_points: the quadrature points of the weak faces, out[point][3] (host), in the order faces ascending /
 *       cells of a face lexicographic with the lower tangential axis fastest / q = q1 + 3 q2 - where the Dirichlet function is
 *       evaluated (velocity.quadrature_point(q), operators.h:1911-1914).
 *   nitsche_rhs: StokesNitscheMatrixFreeOperator::vmult(dst): the boundary functional of the Dirichlet data g (host array
 *       [point][3] in that order) is ADDED to dst_u / dst_p (device).  Synchronises `stream` once (upload of g).
 * Not built: the CIP interior-face term (delta0 != 0, operators.h:1603-1638) and the convection modes (form / jacobian). */
int stfem_stokes_set_weak_boundaries(stfem_stokes_ctx *ctx, int weak_mask, int outflow_mask, double penalty1, double penalty2);
int64_t stfem_stokes_n_face_points(const stfem_stokes_ctx *ctx);
int stfem_stokes_face_points(const stfem_stokes_ctx *ctx, double *out);
int stfem_stokes_nitsche_rhs(stfem_stokes_ctx *ctx, const double *g_at_face_points, double *dst_u, double *dst_p, void *stream);
const char *stfem_stokes_last_hip_error(void);

/* The pressure space by itself - what the solver around the operator needs of it (tests/tp_03stokes.cc:404-425, 1047-1062;
 * include/exact_solution.h:503-649; the pressure transfer of the Stokes multigrid levels, include/stmg.h:557-600).
 *   pressure_ctx: the scalar context behind the pressure vectors, owned by the Stokes context: wrap pressure arrays as its one-block
 *       vectors (stfem_vector_wrap) for the vector arithmetic.  FE_Q(1): a degree-1 context on the mesh without constraints - load
 *       vectors, error norms and space transfers of a FE_Q(1) function work on it too.  FE_DGP(1): a CARRIER with 4 n_cells DoFs (a
 *       degree-1 context on 1 x 1 x (n_cells - 1) cells): vector arithmetic only.
 *   pressure_mean_vectors: host arrays of n_pressure_dofs entries: ones = coefficients of the constant 1, weights = (1, psi_j), and the
 *       volume: mean(p) = weights . p / volume (VectorTools::compute_mean_value / add_constant).  Axis-aligned uniform meshes.
 *   pressure_quadrature_points / pressure_difference: QGauss(nq)^3 on the cells, out[cell][q][3]; { sum JxW (p_h - p)^2, max |p_h - p| }
 *       for exact values at those points (host) and a device pressure array, both pressure spaces (VectorTools::integrate_difference).
 *   dgp_prolongate / dgp_restrict: FE_DGP(1) between a mesh and the mesh with twice the cells per direction (embedding / transpose). */
int stfem_stokes_pressure_ctx(stfem_stokes_ctx *ctx, stfem_ctx **out);
int stfem_stokes_pressure_mean_vectors(stfem_stokes_ctx *ctx, double *ones, double *weights, double *volume);
int stfem_stokes_pressure_quadrature_points(const stfem_stokes_ctx *ctx, int nq, double *out);
int stfem_stokes_pressure_difference(stfem_stokes_ctx *ctx, int nq, const double *p, const double *exact_at_points, double out[2], void *stream);
int stfem_stokes_dgp_prolongate(stfem_stokes_ctx *fine, stfem_stokes_ctx *coarse, double *dst_fine, const double *src_coarse, int add, void *stream);
int stfem_stokes_dgp_restrict(stfem_stokes_ctx *fine, stfem_stokes_ctx *coarse, double *dst_coarse, const double *src_fine, int add, void *stream);

/* PreconditionVanka over a BlockSlice with two variables (reference include/stmg.h:626-738, 832-872, as tests/tp_03stokes.cc:537-540,
 * 714-726 creates it: K_mask empty, M_mask(0, 0) only): per cell the inverse of
 *     B((i, k), (j, l)) = valence(k) * (Alpha(i, j) K_{iv,jv}(k, l) + [iv = jv = velocity] Beta(i, j) M(k, l)),
 * i, j = blocks of the BlockSlice, block_variable[i] = 0 (velocity) / 1 (pressure), K = the assembled Stokes matrix of the context
 * (weak boundary faces included; strong velocity constraints: row and column dropped, diagonal kept), M = the vector mass;
 * vmult / step: dst = (accumulate ? dst : 0) + omega * sum over cells of scatter(B_c^-1 gather(src)), blocks in BlockSlice order
 * (at most 8; velocity blocks 3 * n_velocity_dofs doubles, pressure blocks n_pressure_dofs).  Axis-aligned uniform meshes (<= 27
 * distinct blocks, read off the operator applied to unit vectors); general meshes: STFEM_ERR_UNSUPPORTED.  fp64. */
typedef struct stfem_stokes_vanka stfem_stokes_vanka;
int stfem_stokes_vanka_create(stfem_stokes_ctx *ctx, int n_blocks, const int32_t *block_variable, const double *Alpha, const double *Beta,
                              stfem_stokes_vanka **out);
void stfem_stokes_vanka_destroy(stfem_stokes_vanka *v);
int stfem_stokes_vanka_n_classes(const stfem_stokes_vanka *v);
int stfem_stokes_vanka_vmult(stfem_stokes_vanka *v, double *const *dst_blocks, const double *const *src_blocks, void *stream);
int stfem_stokes_vanka_step(stfem_stokes_vanka *v, double *const *dst_blocks, double omega, int accumulate, const double *const *src_blocks,
                            void *stream);
const char *stfem_stokes_vanka_last_error(void);

const char *stfem_strerror(int status);
/* text of the last failing HIP call on this thread ("" if none) */
const char *stfem_last_hip_error(void);
/* Named trace ranges (roctx, bound at run time: no-ops without the profiler's library or with STFEM_TRACE=0).  The library
 * opens "vmult" / "Tvmult" around stfem_st_vmult and "vanka" around stfem_vanka_vmult - the names of the reference's
 * TimerOutput scopes (operators.h:539, 564, 590; stmg.h:835); callers add their own ("gmg", stmg.h:1335; "step"). */
void stfem_trace_push(const char *name);
void stfem_trace_pop(void);
/* name of the kernel variant the last stfem_st_vmult on this ctx dispatched to (for profiles) */
const char *stfem_last_kernel_name(const stfem_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* STFEM_H */
