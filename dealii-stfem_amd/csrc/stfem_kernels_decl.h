// Declarations of one precision instantiation (included by stfem_kernels.h once with
// STFEM_REAL = double / STFEM_NS = f64 and once with float / f32; no include guard on purpose).
namespace stfem {
namespace STFEM_NS {

using real_t = STFEM_REAL;

struct SweepParams {
  const real_t *src[MAX_BLOCKS];
  real_t *dst[MAX_BLOCKS];
  real_t alpha[MAX_BLOCKS * MAX_BLOCKS]; // [j*nbi + i], already transposed for Tvmult
  real_t beta[MAX_BLOCKS * MAX_BLOCKS];
  int nbi, nbo;         // input (source) and output (destination) temporal blocks
  int ncx, ncy, ncz;    // cells per direction
  int nx, ny, nz;       // DoFs per direction
  int64_t ncells;
  int dmask;            // Dirichlet faces
  real_t vol;           // hx*hy*hz
  real_t ihx2, ihy2, ihz2;
  const real_t *coef_lap;  // per cell or nullptr
  const real_t *coef_mass; // per cell or nullptr
  int experiment;          // ablation bits for cell_core (256: no LDS traffic in the core); results wrong if set
  real_t eo_Si[EO_N], eo_L[EO_N]; // even-odd packed: interpolation (weights folded), 1D Laplacian
  // Cartesian fast path: simultaneous diagonalisation of the 1D nodal mass / stiffness pair
  // (host_tables.h: fd_W, fd_lam), eigenvalues pre-scaled by 1/h_d^2 per direction
  real_t fd_W[EO_N], fd_lx[8], fd_ly[8], fd_lz[8];
  // general-geometry path: plain interpolation S, collocation derivative D and D^T, and the
  // per-quadrature-point metric records [cell][qz][qy][qx][8] = (Gxx,Gxy,Gxz,Gyy,Gyz,Gzz,Mq,pad)
  real_t eo_S[EO_N], eo_Dq[EO_N], eo_DqT[EO_N];
  const real_t *metric;
  // Stokes (csrc/stfem_stokes.hip, Kronecker path): with the three velocity components as the blocks of a FE_Q(2) sweep the kernel
  // can add the pressure gradient term - gscale * B^T p to what it stores (no second kernel reading and writing the velocity
  // again): gp = the FE_Q(1) pressure on the (ncx + 1)(ncy + 1)(ncz + 1) cell vertices (nullptr: off), gw[d][f][a][j] = the 1D
  // integrals of direction d between velocity node a and pressure vertex j, f = 0: derivative form (C), f = 1: value form (h_d N)
  const real_t *gp;
  real_t gw[3][2][3][2];
  real_t gscale;
};

// Decomposition used by the "tile" variant: a workgroup owns a tile of cw x rows cells in x-y and
// marches through lz cell layers in z, accumulating shared DoFs in LDS.  x-neighbouring tiles
// are launched in two colours (odd tiles read-modify-write the shared columns); partial sums on
// a tile's upper y/z faces go to per-tile halo slabs and are added to their owner by a small
// fix-up kernel.  No global atomics and no zeroing of dst are needed.
struct TilePlan {
  int cw, rows;      // cells per tile row (wx waves side by side), cell rows (one wave row each) per tile
  int wx;            // waves per cell row: the workgroup has 64 * wx * rows threads
  int ntx, nty, ntc; // tiles in x, y and chunks in z
  int lz;            // cell layers per chunk (the last chunk may have fewer)
  int tX, tY, zp;    // slab extents: P*cw+1, P*rows+1, P*lz+1
  real_t *yh, *zh;   // halo slabs: yh[tile][block][zl][X], zh[tile][block][Y][X]
  real_t *xl, *xr;   // x-face slabs of odd tiles: [tile][block][zl][Y] (their X = 0 / X = xext columns)
  int add;           // accumulate into dst instead of overwriting
  int xcolor;        // parity of the tile x-index handled by this launch
  int stagger;       // start delay (units of 1024 cycles) of every other group of stagger_div blocks
  int stagger_div;
  int experiment;    // ablation bit mask (STFEM_EXP; results are wrong when nonzero): 1 no src loads,
                     // 2 no cell core, 4 no LDS accumulation, 8 no store phase
  long long *timeline; // diagnostic builds (-DSTFEM_TIMELINE): [block][wave][layer][16] timestamps
};

// Decomposition used by the "pencil" variant (stfem_pencil.hip): every wave owns cpw - 1 cells in x
// (plus the halo slot: its left neighbour's last cell, computed once more) times ty cells in y and
// marches through the lz layers of a z-chunk; four waves stacked in y form a workgroup tile.  All
// pencils run in one launch; partial sums on a workgroup tile's upper y / z faces go to halo slabs
// and are added to their owner by st_pencil_fixup.
struct PencilPlan {
  int cpw, ty;        // cell slots per wave in x (cpw - 1 owned + the halo slot), cell rows per wave in y
  int ntx, ntyw, ntc; // pencils in x, workgroup tiles in y, chunks in z
  int lz;             // cell layers per chunk (longest)
  int zb[65];         // first layer of every chunk, zb[ntc] = ncz (long chunks first, short ones last)
  int tX, tYW, zp;    // slab extents: P*cpw+1, P*ty*4+1, P*lz+1
  real_t *yh, *zh;    // halo slabs: yh[tile][block][zl][X], zh[tile][block][Y][X]
  int add;            // accumulate into dst instead of overwriting
  int *work;          // tile counters, one per XCD (32 ints apart), zeroed before every launch
  int grid;           // workgroups to launch: what the device keeps resident (they pull tiles until none is left)
  long long *timeline; // diagnostic builds (-DSTFEM_PENCIL_TIMELINE): [block][wave][layer][cyl][8] timestamps
};
int pencil_geometry(int p, int nbm, int ty, PencilPlan &plan);
int launch_pencil(int p, const SweepParams &prm, const PencilPlan &plan, void *stream);

// Cartesian (axis-aligned uniform box) meshes, per-cell-constant coefficients.
// Variant "atomic": result scattered with global fp64 atomics into a pre-zeroed dst.
// Returns 0, or -2 if (p, nbm) has no instantiation.
int launch_cart_atomic(int p, const SweepParams &prm, void *stream);
const char *cart_atomic_name(int p, int nbm);

// Forward diagonal of ms*M_c + ls*K_c (reference operators.h:1092-1110), accumulated with fp64
// atomics into a zeroed vector.  Cartesian cells use the 1D diagonals m1[a] = (S^T W S)_aa and
// l1[a] = (S^T D^T W D S)_aa; general cells sum over the quadrature points with the metric.
struct DiagParams {
  real_t *diag;
  int ncx, ncy, ncz, nx, ny, p, dmask;
  real_t ms, ls;          // effective scalings (1 where a coefficient replaces them)
  real_t vol, ihx2, ihy2, ihz2;
  const real_t *coef_lap, *coef_mass; // per cell or nullptr (Cartesian path)
  const real_t *metric;               // general path, coefficients baked in
  real_t m1[8], l1[8];                // 1D diagonals (Cartesian)
  real_t S[64], D[64];                // plain S[q][a], D[q][a] (general)
};
int launch_diagonal(const DiagParams &prm, void *stream);

// Variant "tile" (default): fills plan.cw/rows/tX/tY for (p, nbm) and the Cartesian (general = 0) or
// general-geometry kernel; returns 0 or -2.
int tile_geometry(int p, int nbm, int general, TilePlan &plan);
int launch_cart_tile(int p, const SweepParams &prm, const TilePlan &plan, void *stream);
// workgroups of the sweep kernel that fit one CU according to the runtime (0 if unknown)
int tile_occupancy(int p, int nbm, int general);
// fills metric[cell][q][8] from the vertex grid (device pointers); coef_* may be null,
// layout 1 = per cell, 2 = per (cell, q)
int launch_build_metric(int p, const int nc[3], const double *d_vertices, const double *d_xq,
                        const double *d_wq, const real_t *coef_lap, int lap_layout,
                        const real_t *coef_mass, int mass_layout, real_t *d_metric, void *stream);
const char *cart_tile_name(int p, int nbm);

} // namespace STFEM_NS
} // namespace stfem
