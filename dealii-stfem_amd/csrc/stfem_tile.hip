// "Tile" variant of the fused space-time cell sweep (Cartesian meshes), the default path.
//
// A 256-thread workgroup (4 waves) owns a tile of CW x 4 cells in x-y (one wave per cell row)
// and marches through LZ cell layers in z:
//
//   per layer:  gather src planes from HBM (5 contiguous doubles per row, faces shared through
//               L1/L2; the next layer is prefetched into registers during the compute)
//               -> cell_core (registers + wave-private LDS transposes)
//               -> accumulate the (p+1)^3 results of all cells of the layer in an LDS slab
//                  (owner lane writes, the other sharers ds_add_f64)
//               -> stream the finished DoF planes of the layer to HBM as rows of p*CW+1
//                  contiguous doubles; the top plane is carried to the next layer in LDS.
//
// Shared DoFs between tiles:
//   x faces: tiles are 2-coloured by the parity of their x index; odd tiles run first and put
//            their partial sums of the two shared columns into contiguous x-slabs; even tiles
//            run second, prefetch the neighbours' slabs at the start of a layer, add them into
//            the LDS slab and store complete rows (no strided plane access, no RMW of dst);
//   y/z faces: partial sums of a tile's upper y / z face go to per-tile halo slabs (plain,
//            contiguous stores) and are added to the owner's rows by st_tile_fixup.
// Every DoF of dst is written by plain stores: no global atomics, no memset of dst.
//
// Replaces the scatter of MatrixFreeOperator::do_cell_integral_range
// (reference include/operators.h:1112-1133, distribute_local_to_global) and the dst = 0 /
// dst.add(...) traffic of SystemMatrix::vmult (operators.h:536-559).
#include "stfem_core.h"

#include <cstdio>
#include <cstdlib>

#ifndef STFEM_TILE_P
#define STFEM_TILE_P 0
#endif

namespace stfem {
namespace STFEM_PREC {

namespace {

// cells of one temporal block set a wave holds (the tile row must fit the 64 lanes of a store instruction)
constexpr int tile_cells_per_wave(int p, int nbm)
{
  const int cpw = (64 / (p + 1)) / nbm;
  return p * cpw + 1 > 64 ? 63 / p : cpw;
}
// Cell groups a wave handles one after the other (rows of two wave-widths).  Round 1 ran the fp64 Cartesian
// kernels with SX = 2 (7 % faster on cfg 1); since round 3 the carried plane and the x-slab values have LDS
// regions of their own (fixed summation order, see the kernel), with which an SX = 2 slab no longer fits twice
// into a CU - and the Cartesian systems are served by the pencil sweep (stfem_pencil.hip) anyway.
constexpr int tile_wx(int, int) { return 1; }
constexpr int tile_sx(int, int, bool) { return 1; }

constexpr int tile_threads(int p, int nbm) { return 256 * tile_wx(p, nbm); }
constexpr int tile_min_blocks(int p, int nbm, int minw) { return minw / tile_wx(p, nbm) > 0 ? minw / tile_wx(p, nbm) : 1; }

template <int P, int NBM, bool GEN> struct TileGeom {
  using G = Geometry<P, NBM>;
  static constexpr int N = P + 1;
  static constexpr int CWW = tile_cells_per_wave(P, NBM); // cells per wave
  static constexpr int WX = tile_wx(P, NBM);              // waves per cell row
  static constexpr int SX = tile_sx(P, NBM, GEN);              // cell groups a wave handles in turn
  static constexpr int ROWS = 4;                           // cell rows (waves along y)
  static constexpr int NWAVES = WX * ROWS;
  static constexpr int NT = 64 * NWAVES;                   // threads per workgroup
  static constexpr int CW = WX * SX * CWW;                 // cells per tile row
  static constexpr int TX = P * CW + 1;
  static constexpr int LPR = TX <= 32 ? 32 : 64;           // store phase: lanes per slab row
  static constexpr int RPI = 64 / LPR;                     // rows per wave-instruction
  static constexpr int RPP = NWAVES * RPI;                 // rows per pass of the workgroup
  static constexpr int TY = P * ROWS + 1;
  static constexpr int PLANE = TX * TY;
  static constexpr int ACC = NBM * N * PLANE;              // accumulation slab (aliases trans)
  static constexpr int LDS_PER_WAVE = CWW * NBM * G::CBS;  // transpose slab of one wave
  static constexpr int TRANS = NWAVES * LDS_PER_WAVE;      // transpose slabs
  static constexpr int MAIN = ACC > TRANS ? ACC : TRANS;
  static constexpr int CARRY = NBM * PLANE;                // top plane carried between layers: its own LDS region
  static constexpr int CARRY_REGS = (CARRY + NT - 1) / NT; // copy passes per thread
  static constexpr int XSL = 2 * NBM * N * TY;             // x-slab values of the layer (both sides): their own region
  static constexpr int LDS_DOUBLES = MAIN + CARRY + XSL;
};

struct TileCoords {
  int tx, ty, tc;     // tile indices
  int cx0, cy0, cz0;  // first cell
  int ncx, ncy, nlay; // active cells / layers in this tile
  bool last_x, last_y, last_z;
};

// z-chunks are as equal as possible: chunk c covers cell layers [c*ncz/ntc, (c+1)*ncz/ntc)
__device__ __forceinline__ int chunk_begin(int c, int ncz, int ntc) { return int(int64_t(c) * ncz / ntc); }

__device__ __forceinline__ TileCoords tile_coords(const SweepParams &prm, const TilePlan &tp, int tx, int ty, int tc)
{
  TileCoords t;
  t.tx = tx;
  t.ty = ty;
  t.tc = tc;
  t.cx0 = t.tx * tp.cw;
  t.cy0 = t.ty * tp.rows;
  t.cz0 = chunk_begin(t.tc, prm.ncz, tp.ntc);
  t.ncx = min(tp.cw, prm.ncx - t.cx0);
  t.ncy = min(tp.rows, prm.ncy - t.cy0);
  t.nlay = chunk_begin(t.tc + 1, prm.ncz, tp.ntc) - t.cz0;
  t.last_x = t.tx == tp.ntx - 1;
  t.last_y = t.ty == tp.nty - 1;
  t.last_z = t.tc == tp.ntc - 1;
  return t;
}

// XCD-aware numbering: blocks b, b+8, b+16, ... share an XCD (round-robin dispatch), so give
// every XCD one contiguous range of tiles (neighbouring tiles then share src faces in one L2).
__device__ __forceinline__ int logical_block(int b, int nblocks)
{
  const int per = nblocks / 8, rem = nblocks % 8;
  const int xcd = b % 8, slot = b / 8;
  return xcd * per + min(xcd, rem) + slot;
}

template <int P>
__device__ __forceinline__ void load_plane(const real_t *__restrict__ s, int nx, real_t (&PA)[(P + 1) * (P + 1)])
{
  constexpr int N = P + 1;
  STFEM_UNROLL
  for (int y = 0; y < N; ++y)
    STFEM_UNROLL
  for (int x = 0; x < N; ++x) PA[y * N + x] = s[int64_t(y) * nx + x];
}

// The same gather issued through inline asm: the compiler does not track these loads, so it
// will not drain them (and, vmcnt being in-order, every store issued after them) with a
// vmcnt(0) at the next use.  The caller waits with wait_vmcnt(n), n <= number of VMEM
// instructions this wave has issued since, before touching PA.
// ASYNC = false falls back to ordinary loads: REQUIRED for every instantiation whose register
// allocation spills, because the compiler would store a spilled destination register to scratch
// right after the asm statement, i.e. before the load it does not know about has landed.  Only
// the fp64 Cartesian kernels with up to three blocks (no spills, checked with tools/kinfo.sh) use it.
template <int P, bool ASYNC>
__device__ __forceinline__ void load_plane_async(const real_t *s, int nx, real_t (&PA)[(P + 1) * (P + 1)])
{
  constexpr int N = P + 1;
#ifdef STFEM_F32
  load_plane<P>(s, nx, PA); // fp32: ordinary (compiler-tracked) loads
  return;
#else
  if (!ASYNC) {
    load_plane<P>(s, nx, PA);
    return;
  }
  typedef double d2 __attribute__((ext_vector_type(2)));
  STFEM_UNROLL
  for (int y = 0; y < N; ++y) {
    const real_t *row = s + int64_t(y) * nx;
    STFEM_UNROLL
    for (int x = 0; x + 1 < N; x += 2) {
      d2 v;
      asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&v"(v) : "v"(row), "n"(x * 8) : "memory");
      PA[y * N + x] = v.x;
      PA[y * N + x + 1] = v.y;
    }
    if (N & 1) {
      real_t v;
      asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=&v"(v) : "v"(row), "n"((N - 1) * 8) : "memory");
      PA[y * N + N - 1] = v;
    }
  }
#endif
}

template <int CNT> __device__ __forceinline__ void wait_vmcnt_imm()
{
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory");
}
// waits until at most n (wave-uniform, clamped to [0, 32]) VMEM operations are outstanding
__device__ __forceinline__ void wait_vmcnt(int n)
{
  n = __builtin_amdgcn_readfirstlane(n);
  if (n >= 32) wait_vmcnt_imm<32>();
  else if (n >= 24) wait_vmcnt_imm<24>();
  else if (n >= 20) wait_vmcnt_imm<20>();
  else if (n >= 16) wait_vmcnt_imm<16>();
  else if (n >= 12) wait_vmcnt_imm<12>();
  else if (n >= 8) wait_vmcnt_imm<8>();
  else if (n >= 4) wait_vmcnt_imm<4>();
  else wait_vmcnt_imm<0>();
}

// waves per SIMD the general-geometry kernel is compiled for (launch bounds)
#ifndef STFEM_GEN_WAVES
#define STFEM_GEN_WAVES 1
#endif

// (non-temporal dst stores were measured in round 1: no effect)
#define STFEM_DST_STORE(ptr, val) (*(ptr) = (val))

template <int P, int NBM, int MINW, bool ADD, bool COEF, bool GEN, int COLOR>
__global__ __launch_bounds__(tile_threads(P, NBM), tile_min_blocks(P, NBM, MINW))
void st_sweep_cart_tile(const SweepParams prm, const TilePlan tp)
{
  using TG = TileGeom<P, NBM, GEN>;
  using G = Geometry<P, NBM>;
  constexpr int N = TG::N;
  constexpr int NT = TG::NT, LPR = TG::LPR, RPI = TG::RPI, RPP = TG::RPP;
  constexpr bool ASYNC_LOADS = !GEN && NBM <= 3; // see load_plane_async
  constexpr int TX = TG::TX, TY = TG::TY, PLANE = TG::PLANE;
  __shared__ real_t smem[TG::LDS_DOUBLES];
  real_t *acc = smem; // [blk][k][Y][X], aliases the transpose slabs
  // Every DoF of the slab is summed in a FIXED order (bitwise reproducible results): its owner lane stores
  // own + carried plane + x-slab value, then the x-neighbour cell (same wave: LDS operations of a wave execute
  // in order) adds, and after the next barrier the cells of the wave below add theirs (one wave, program order).
  real_t *cr = smem + TG::MAIN;       // [blk][Y][X]: top plane of the previous layer
  real_t *xr = cr + TG::CARRY;        // [side][blk][k][Y]: partial sums of the odd x-neighbours' shared columns

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  real_t *lds = smem + wave * TG::LDS_PER_WAVE;

  // tiles of this launch's x colour
  const int ntxh = (tp.ntx - COLOR + 1) / 2; // tiles of this launch's x colour
  const int nblocks = ntxh * tp.nty * tp.ntc;
  const int id = logical_block(blockIdx.x, nblocks);
  const TileCoords t =
    tile_coords(prm, tp, 2 * (id % ntxh) + COLOR, (id / ntxh) % tp.nty, id / (ntxh * tp.nty));
  const int tile_id = t.tx + tp.ntx * (t.ty + tp.nty * t.tc);

  const bool lane_ok = lane < TG::CWW * NBM * N;
  const int l = lane_ok ? lane : 0;
  const int k = l % N;
  const int blk = (l / N) % NBM;
  constexpr int SX = TG::SX;
  const int cxw = l / (N * NBM); // cell within the wave's group
  const int cyl = wave / TG::WX;
  // per cell group h of this wave
  int cxl[SX], cx[SX];
  bool cell_ok[SX], in_active[SX], out_active[SX], own_x_hi[SX];
  int64_t cell_xy[SX];
  const int cy = t.cy0 + ((lane_ok && cyl < t.ncy) ? cyl : 0);
  STFEM_UNROLL
  for (int h = 0; h < SX; ++h) {
    cxl[h] = ((wave % TG::WX) * SX + h) * TG::CWW + cxw; // cell within the tile row
    cell_ok[h] = lane_ok && cxl[h] < t.ncx && cyl < t.ncy;
    cx[h] = t.cx0 + (cell_ok[h] ? cxl[h] : 0);
    in_active[h] = cell_ok[h] && blk < prm.nbi;
    out_active[h] = cell_ok[h] && blk < prm.nbo;
    cell_xy[h] = cx[h] + int64_t(prm.ncx) * cy;
    own_x_hi[h] = cxl[h] == t.ncx - 1; // last active cell of the row owns its x = P column
  }
  const int64_t cells_per_layer = int64_t(prm.ncx) * prm.ncy;

  real_t aK0[NBM], aM0[NBM];
  STFEM_UNROLL
  for (int i = 0; i < NBM; ++i) {
    const bool ok = blk < prm.nbo && i < prm.nbi;
    aK0[i] = ok ? prm.alpha[blk * prm.nbi + i] * prm.vol : real_t(0);
    aM0[i] = ok ? prm.beta[blk * prm.nbi + i] * prm.vol : real_t(0);
  }

  // which entries of this lane's result plane it initialises in the LDS slab ("owner")
  const bool own_y_hi = cyl == t.ncy - 1;

  // Dirichlet rows only matter for tiles that touch the domain boundary (uniform test)
  const bool xy_boundary = ((prm.dmask & 1) && t.tx == 0) || ((prm.dmask & 2) && t.last_x) ||
                           ((prm.dmask & 4) && t.ty == 0) || ((prm.dmask & 8) && t.last_y);

  const int64_t plane_stride = int64_t(prm.nx) * prm.ny;
  // lanes that feed nothing still load from a valid address; their planes are never used
  const real_t *src_lane[SX];
  int a_off[SX]; // this lane's result plane in the accumulation slab
  STFEM_UNROLL
  for (int h = 0; h < SX; ++h) {
    src_lane[h] = prm.src[in_active[h] ? blk : 0] + int64_t(P) * cx[h] + int64_t(prm.nx) * (int64_t(P) * cy) + plane_stride * k;
    a_off[h] = ((blk * N + k) * TY + P * cyl) * TX + P * cxl[h];
  }

  const int xext = P * t.ncx, yext = P * t.ncy;
  const int ymax = t.last_y ? yext + 1 : yext; // rows [0, ymax) go to dst, row yext to yh otherwise
  // x faces: odd tiles divert their shared columns to the x-slabs, even tiles collect them
  constexpr bool odd = COLOR == 1;
  const bool fast_tile = t.ncx == TG::CW && t.ncy == TG::ROWS && !t.last_y; // wave-uniform
  const bool collect_left = !odd && t.tx > 0, collect_right = !odd && !t.last_x;
  constexpr int XE = (2 * NBM * N * TY + NT - 1) / NT; // slab values per thread and layer
  const int nrows = prm.nbo * N * TY;
  const int64_t tile_goff = int64_t(P) * t.cx0 + int64_t(prm.nx) * (int64_t(P) * t.cy0) +
                            plane_stride * (int64_t(P) * t.cz0);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);

  real_t PA[SX][N * N];
  STFEM_UNROLL
  for (int h = 0; h < SX; ++h) load_plane_async<P, ASYNC_LOADS>(src_lane[h] + plane_stride * (int64_t(P) * t.cz0), prm.nx, PA[h]);
#ifndef STFEM_F32
  if (ASYNC_LOADS) wait_vmcnt_imm<0>();
#endif
#ifdef STFEM_ABLATION
#define STFEM_LAYER_BARRIER() do { if (!(ex & 2048)) __syncthreads(); } while (0)
  const int ex = tp.experiment; // timing experiments only (tools/ablate.sh); results are wrong
#else
#define STFEM_LAYER_BARRIER() __syncthreads()
  constexpr int ex = 0;
#endif
  // stagger: the two workgroups of a CU start together with identical work and would otherwise
  // run their compute and their memory phases in lockstep; delaying the one in the odd wave slot
  // lets one stream while the other computes
  if (tp.stagger > 0) {
    // key: which of the co-resident workgroups of a CU this is (first round of the dispatch)
    if ((blockIdx.x / tp.stagger_div) & 1)
      for (int i = 0; i < tp.stagger; ++i) __builtin_amdgcn_s_sleep(16); // 16 x 64 cycles each
  }

#ifdef STFEM_TIMELINE
  // phase timestamps of the even-colour launch (constant 100 MHz clock, comparable across CUs);
  // slot 15 of layer 0 holds HW_ID | XCC_ID << 32
#define STFEM_TL(i)                                                                              \
  do {                                                                                           \
    if (COLOR == 0 && tp.timeline && lane == 0)                                                  \
      tp.timeline[((int64_t(blockIdx.x) * TG::NWAVES + wave) * tp.lz + layer) * 16 + (i)] = wall_clock64(); \
  } while (0)
  if (COLOR == 0 && tp.timeline && lane == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    tp.timeline[((int64_t(blockIdx.x) * TG::NWAVES + wave) * tp.lz) * 16 + 15] = (long long)hw | ((long long)xcc << 32);
  }
#else
#define STFEM_TL(i) do {} while (0)
#endif
  real_t dummy = real_t(0);
  for (int layer = 0; layer < t.nlay; ++layer) {
    STFEM_TL(0);
    const int cz = t.cz0 + layer;
    const bool last_layer = layer == t.nlay - 1;
    const bool z_boundary = ((prm.dmask & 16) && cz == 0) || ((prm.dmask & 32) && cz == prm.ncz - 1);
    const bool masked = xy_boundary || z_boundary;
    PlaneMask pm[SX];
    STFEM_UNROLL
    for (int h = 0; h < SX; ++h) {
      pm[h] = plane_mask<P>(prm, cx[h], cy, cz, k);
      if (masked) {
        STFEM_UNROLL
        for (int y = 0; y < N; ++y)
          STFEM_UNROLL
        for (int x = 0; x < N; ++x)
          if (constrained<P>(pm[h], y, x)) PA[h][y * N + x] = real_t(0);
      }
    }

    // even tiles: fetch the odd neighbours' partial sums of the shared columns for this layer
    // (issued before the compute, consumed after it)
    real_t xe[XE];
    STFEM_UNROLL
    for (int m = 0; m < XE; ++m) xe[m] = real_t(0);
    // slab entry m of this thread: (side, row) -> LDS slab index or -1; evaluated twice (here for
    // the load, after the core for the add) rather than kept in registers across the core
    auto xe_slot = [&](int m, int &side, int &j, int &kk, int &Y) -> int {
      const int e = tid + NT * m;
      side = e >= nrows ? 1 : 0;
      const int row = e - side * nrows;
      Y = row % TY;
      const int jk = row / TY;
      kk = jk % N;
      j = jk / N;
      const bool to_dst = Y < ymax && (kk < P || (last_layer && t.last_z));
      const bool ok = e < 2 * nrows && to_dst && (side == 0 ? collect_left : collect_right);
      return ok ? row * TX + (side == 0 ? 0 : xext) : -1;
    };
    if (collect_left || collect_right) {
      STFEM_UNROLL
      for (int m = 0; m < XE; ++m) {
        int side, j, kk, Y;
        if (xe_slot(m, side, j, kk, Y) >= 0) {
          const int nid = tile_id + (side == 0 ? -1 : 1);
          const real_t *slab = (side == 0 ? tp.xr : tp.xl) + int64_t(nid) * NBM * tp.zp * tp.tY;
          xe[m] = slab[(j * tp.zp + P * layer + kk) * tp.tY + Y];
        }
      }
    }

    STFEM_TL(1);
    STFEM_UNROLL
    for (int h = 0; h < SX; ++h) {
      real_t aK[NBM], aM[NBM];
      if (COEF) { // per-cell coefficients (operators.h:1060-1087) folded into the temporal weights
        const int64_t c = cell_xy[h] + cells_per_layer * cz;
        const real_t fK = prm.coef_lap ? prm.coef_lap[c] : real_t(1);
        const real_t fM = prm.coef_mass ? prm.coef_mass[c] : real_t(1);
        STFEM_UNROLL
        for (int i = 0; i < NBM; ++i) {
          aK[i] = aK0[i] * fK;
          aM[i] = aM0[i] * fM;
        }
      } else {
        STFEM_UNROLL
        for (int i = 0; i < NBM; ++i) {
          aK[i] = aK0[i];
          aM[i] = aM0[i];
        }
      }
      if (GEN)
        cell_core_general<P, NBM>(prm, lds, cxw, blk, k, in_active[h], out_active[h], aK, aM,
                                  prm.metric + (cell_xy[h] + cells_per_layer * cz) * (8 * N * N * N), PA[h]);
      else if (!(ex & 2))
        cell_core<P, NBM>(prm, lds, cxw, blk, k, in_active[h], out_active[h], aK, aM, PA[h]);
      if (SX > 1) pin(PA[h]); // finish this group before the next one starts
    }
    // the slab values have long arrived; consuming them here on every path keeps the compiler
    // from draining the src prefetch (issued below) when their registers are recycled later
    STFEM_UNROLL
    for (int m = 0; m < XE; ++m) asm volatile("" : "+v"(xe[m]));
    if (collect_left || collect_right) { // (zeros where nothing is collected; read by the owner lanes after the barrier)
      STFEM_UNROLL
      for (int m = 0; m < XE; ++m)
        if (tid + NT * m < 2 * nrows) xr[tid + NT * m] = xe[m];
    }

    if (masked) {
      STFEM_UNROLL
      for (int h = 0; h < SX; ++h)
        STFEM_UNROLL
      for (int y = 0; y < N; ++y)
        STFEM_UNROLL
      for (int x = 0; x < N; ++x)
        if (constrained<P>(pm[h], y, x)) PA[h][y * N + x] = real_t(0);
    }

    STFEM_TL(2);
    STFEM_LAYER_BARRIER(); // all waves are done with the transpose slabs: the region becomes `acc`
    STFEM_TL(3);

    // owner lanes initialise their DoFs: own value + plane carried from the previous layer + x-slab value
    STFEM_UNROLL
    for (int h = 0; h < SX; ++h)
      if (out_active[h] && !(ex & 4)) {
        real_t *a = acc + a_off[h];
        const real_t *c0 = cr + blk * PLANE + (P * cyl) * TX + P * cxl[h];
        const real_t *x0 = xr + (blk * N + k) * TY + P * cyl;
        const bool use_c = layer > 0 && k == 0;
        const bool use_l = collect_left && cxl[h] == 0, use_r = collect_right && own_x_hi[h];
        STFEM_UNROLL
        for (int y = 0; y < N; ++y)
          STFEM_UNROLL
        for (int x = 0; x < N; ++x) {
          const bool owned = (x < P || own_x_hi[h]) && (y < P || own_y_hi);
          if (!owned) continue;
          real_t v = PA[h][y * N + x];
          if (use_c) v += c0[y * TX + x];
          if (x == 0 && use_l) v += x0[y];
          if (x == P && use_r) v += x0[nrows + y];
          a[y * TX + x] = v;
        }
      }
    // ... the x = P column of a cell is the x = 0 column of the next cell of the row, which this same wave owns
    STFEM_UNROLL
    for (int h = 0; h < SX; ++h)
      if (out_active[h] && !(ex & 4) && !own_x_hi[h]) {
        real_t *a = acc + a_off[h];
        STFEM_UNROLL
        for (int y = 0; y < N; ++y)
          if (y < P || own_y_hi) atomicAdd(&a[y * TX + P], PA[h][y * N + P]);
      }
    STFEM_TL(4);
    __syncthreads();
    STFEM_TL(5);
    // ... and the y = P row that of the cells above, owned by the next wave
    if (!own_y_hi) {
      STFEM_UNROLL
      for (int h = 0; h < SX; ++h)
        if (out_active[h] && !(ex & 4)) {
          real_t *a = acc + a_off[h];
          STFEM_UNROLL
          for (int x = 0; x < N; ++x) atomicAdd(&a[P * TX + x], PA[h][P * N + x]);
        }
    }
    // the result planes are in LDS now: fetch the next layer's src planes (in flight during the
    // store phase; a separate prefetch buffer one layer ahead would need > 256 VGPRs)
    if (!last_layer && !(ex & 1)) {
      STFEM_UNROLL
      for (int h = 0; h < SX; ++h) load_plane_async<P, ASYNC_LOADS>(src_lane[h] + plane_stride * (int64_t(P) * (cz + 1)), prm.nx, PA[h]);
    }
    STFEM_TL(6);
    STFEM_LAYER_BARRIER();
    STFEM_TL(7);

    // stream the finished planes to their destination: k = 0..P-1, and k = P on the last layer.
    // Fully unrolled with predicates: with loops here the compiler drains the prefetch loads
    // above (s_waitcnt vmcnt(0)) before the first store.
    const int kend = last_layer ? N : P;
    if (fast_tile && !(ex & 8)) {
      // Full interior tile (the common case): all extents are compile-time constants, so the
      // store phase is straight-line code: rows [0, P*ROWS) of every finished plane go to dst,
      // row P*ROWS to the y-halo slab.
      constexpr int XEXT = P * TG::CW, YMAX = P * TG::ROWS;
      constexpr int NOF = (YMAX + RPP - 1) / RPP;
      int lane_s = lane;
      asm volatile("" : "+v"(lane_s));
      const int hw = RPI * wave_u + lane_s / LPR, X = lane_s % LPR;
      const bool x_lane = X <= XEXT;
      const bool divert_lane = odd && (X == 0 || (X == XEXT && !t.last_x));
      const unsigned lane_goff = X + prm.nx * (lane_s / LPR), lane_zoff = X + tp.tX * (lane_s / LPR);
      const int lane_aoff = hw * TX + X;
      real_t *const xslab_out = (X == 0 ? tp.xl : tp.xr) + int64_t(tile_id) * NBM * tp.zp * tp.tY;
      STFEM_UNROLL
      for (int j = 0; j < NBM; ++j) {
        if (j >= prm.nbo) continue;
        // all LDS reads of this block first (unconditionally: every index stays inside the slab),
        // then all stores under ONE lane predicate
        real_t sv[N][NOF];
        STFEM_UNROLL
        for (int kk = 0; kk < N; ++kk)
          STFEM_UNROLL
        for (int o = 0; o < NOF; ++o) sv[kk][o] = acc[(j * N + kk) * PLANE + lane_aoff + o * RPP * TX];
        real_t *dj = prm.dst[j] + tile_goff + int64_t(prm.nx) * (RPI * wave_u) + plane_stride * (int64_t(P) * layer);
        const bool main_lane = x_lane && !(odd && divert_lane);
        if (main_lane) {
          STFEM_UNROLL
          for (int kk = 0; kk < P; ++kk)
            STFEM_UNROLL
          for (int o = 0; o < NOF; ++o) {
            if ((YMAX % RPP != 0) && hw + RPP * o >= YMAX) continue;
            real_t *d = dj + plane_stride * kk + int64_t(o * RPP) * prm.nx + lane_goff;
            if (ADD) *d += sv[kk][o];
            else STFEM_DST_STORE(d, sv[kk][o]);
          }
        }
        if (odd && divert_lane) { // the two shared columns of an odd tile go to its x-slabs
          real_t *xs = xslab_out + (j * tp.zp + P * layer) * tp.tY;
          STFEM_UNROLL
          for (int kk = 0; kk < P; ++kk)
            STFEM_UNROLL
          for (int o = 0; o < NOF; ++o) {
            const int Y = hw + RPP * o;
            if ((YMAX % RPP == 0) || Y < YMAX) xs[kk * tp.tY + Y] = sv[kk][o];
          }
        }
        if (last_layer) { // the top plane leaves too: to the z-halo slab, or to dst on the last chunk
          real_t *zj = tp.zh + (int64_t(tile_id) * NBM + j) * tp.tY * tp.tX + tp.tX * (RPI * wave_u);
          real_t *xs = xslab_out + (j * tp.zp + P * layer) * tp.tY;
          STFEM_UNROLL
          for (int o = 0; o < NOF; ++o) {
            const int Y = hw + RPP * o;
            const bool row_ok = (YMAX % RPP == 0) || Y < YMAX;
            if (row_ok && x_lane) {
              real_t *d = dj + plane_stride * P + int64_t(o * RPP) * prm.nx + lane_goff;
              if (!t.last_z) (zj + o * RPP * tp.tX)[lane_zoff] = sv[P][o];
              else if (odd && divert_lane) xs[P * tp.tY + Y] = sv[P][o];
              else if (ADD) *d += sv[P][o];
              else STFEM_DST_STORE(d, sv[P][o]);
            }
          }
        }
      }
      // row Y = P*ROWS of every finished plane: to the y-halo slab (the tile is not last in y)
      STFEM_UNROLL
      for (int o = 0; o < (NBM * N + RPP - 1) / RPP; ++o) {
        const int r = hw + RPP * o;
        const int j = r / kend, kk = r - j * kend;
        if (r < prm.nbo * kend && x_lane)
          tp.yh[((int64_t(tile_id) * NBM + j) * tp.zp + P * layer + kk) * tp.tX + X] =
            acc[((j * N + kk) * TY + YMAX) * TX + X];
      }
    } else
    if (!(ex & 8)) {
      constexpr int NO = (TY + RPP - 1) / RPP; // row passes per plane
      // store-phase lane roles: one slab row per half-wave (32 lanes, X = lane within the half).
      // Derived from a laundered lane id so that they are recomputed here instead of being kept
      // in ~15 VGPRs across the register-critical core.
      int lane_s = lane;
      asm volatile("" : "+v"(lane_s));
      const int hw = RPI * wave_u + lane_s / LPR, X = lane_s % LPR;
      const bool x_lane = X <= xext;
      const bool divert_lane = odd && (X == 0 || (X == xext && !t.last_x));
      real_t *const xslab_out = (X == 0 ? tp.xl : tp.xr) + int64_t(tile_id) * NBM * tp.zp * tp.tY;
      const unsigned lane_goff = X + prm.nx * (lane_s / LPR), lane_zoff = X + tp.tX * (lane_s / LPR);
      const int lane_aoff = hw * TX + X;
      STFEM_UNROLL
      for (int j = 0; j < NBM; ++j) {
        if (j >= prm.nbo) continue;
        // all LDS reads of this block first, then all stores: a VMEM store keeps its address and
        // data VGPRs locked until it has completed (vmcnt), so they must not be recycled
        // from one store to the next
        real_t sv[N][NO];
        STFEM_UNROLL
        for (int kk = 0; kk < N; ++kk)
          STFEM_UNROLL
        for (int o = 0; o < NO; ++o)
          sv[kk][o] = (kk < kend && hw + RPP * o < ymax && x_lane && !(ex & 32)) ? acc[(j * N + kk) * PLANE + lane_aoff + o * RPP * TX] : real_t(0);
        real_t *dj = prm.dst[j] + tile_goff + int64_t(prm.nx) * (RPI * wave_u); // wave-uniform
        real_t *zj = tp.zh + (int64_t(tile_id) * NBM + j) * tp.tY * tp.tX + tp.tX * (RPI * wave_u);
        STFEM_UNROLL
        for (int kk = 0; kk < N; ++kk) {
          if (kk >= kend) continue;
          const int zl = P * layer + kk; // chunk-local plane
          const bool to_zh = kk == P && !t.last_z; // top plane of an inner chunk: z-halo slab
          real_t *xs = xslab_out + (j * tp.zp + zl) * tp.tY;
          STFEM_UNROLL
          for (int o = 0; o < NO; ++o) {
            const int Y = hw + RPP * o;
            if (ex & 64) { dummy += sv[kk][o]; continue; }
            if (Y < ymax && x_lane) {
              const real_t v = sv[kk][o];
              if (to_zh) (zj + o * RPP * tp.tX)[lane_zoff] = v;
              else if (divert_lane) xs[Y] = v;
              else if (ADD) (dj + plane_stride * zl + int64_t(o * RPP) * prm.nx)[lane_goff] += v;
              else STFEM_DST_STORE(&(dj + plane_stride * zl + int64_t(o * RPP) * prm.nx)[lane_goff], v);
            }
          }
        }
      }
      if (!t.last_y && !(ex & 128)) { // row Y = yext of every finished plane: to the y-halo slab
        STFEM_UNROLL
        for (int o = 0; o < (NBM * N + RPP - 1) / RPP; ++o) {
          const int r = hw + RPP * o;
          const int j = r / kend, kk = r - j * kend;
          if (r < prm.nbo * kend && x_lane)
            tp.yh[((int64_t(tile_id) * NBM + j) * tp.zp + P * layer + kk) * tp.tX + X] =
              acc[((j * N + kk) * TY + yext) * TX + X];
        }
      }
    }
    STFEM_TL(8);
    if (!last_layer) { // top plane: carried to the next layer
      STFEM_UNROLL
      for (int m = 0; m < TG::CARRY_REGS; ++m) {
        const int e = tid + NT * m, j = e / PLANE;
        if (e < prm.nbo * PLANE) cr[e] = acc[(j * (N - 1) + P) * PLANE + e];
      }
    }
    STFEM_TL(9);
    STFEM_LAYER_BARRIER(); // slab free again for the next layer's transposes
    STFEM_TL(10);
    // the prefetched planes must have landed before PA is touched; the stores issued after
    // them may stay in flight.  Lower bound of the stores this wave has issued since: one per
    // (block, plane, row group) whose first half-wave row exists.
    {
      int n_o = 0;
      STFEM_UNROLL
      for (int o = 0; o < (TY + RPP - 1) / RPP; ++o) n_o += (RPI * wave_u + RPP * o < ymax) ? 1 : 0;
#if !defined(STFEM_F32)
      if (ASYNC_LOADS) wait_vmcnt((ADD || (ex & 8) || (ex & 64)) ? 0 : prm.nbo * kend * n_o);
#else
      (void)n_o;
#endif
    }
    STFEM_TL(11);
  }
#ifdef STFEM_ABLATION
  if (dummy == real_t(1.2345e30)) tp.zh[0] = dummy; // experiment sink, never true
#endif
}

// Adds the halo partial sums of the lower y / z neighbours to the rows a tile owns on its y = 0
// and z = 0 faces.  One workgroup per tile; 32 lanes per row (contiguous in x), 8 rows at a time,
// no integer divisions in the loops.
template <int P>
__global__ __launch_bounds__(256) void st_tile_fixup(const SweepParams prm, const TilePlan tp, int nbm)
{
  const int id = blockIdx.x;
  const TileCoords t = tile_coords(prm, tp, id % tp.ntx, (id / tp.ntx) % tp.nty, id / (tp.ntx * tp.nty));
  const int has_x = t.tx > 0, has_y = t.ty > 0, has_z = t.tc > 0;
  if (!(has_y | has_z)) return;
  // owned local extents
  const int Xn = P * t.ncx + (t.last_x ? 1 : 0), Yn = P * t.ncy + (t.last_y ? 1 : 0),
            Zn = P * t.nlay + (t.last_z ? 1 : 0);
  const int lpr = tp.tX <= 32 ? 32 : 64, nrg = 256 / lpr; // lanes per row, rows per pass
  const int X = threadIdx.x % lpr, rg = threadIdx.x / lpr;
  if (X >= Xn) return;
  const int64_t plane_stride = int64_t(prm.nx) * prm.ny;
  const int64_t g0 = int64_t(P) * t.cx0 + X + int64_t(prm.nx) * (int64_t(P) * t.cy0) +
                     plane_stride * (int64_t(P) * t.cz0);
  const int dxn = (X == 0) ? has_x : 0; // the x = 0 column also collects from the tiles at tx - 1
  const int XpL = P * tp.cw;            // that column in the left neighbour's coordinates
  const int tid_y = id - tp.ntx, tid_z = id - tp.ntx * tp.nty, tid_yz = tid_z - tp.ntx;
  // top plane of the chunk below, in its own slab coordinates
  const int top_below = has_z ? P * (t.cz0 - chunk_begin(t.tc - 1, prm.ncz, tp.ntc)) : 0;
  { // one workgroup per (tile, output block): the blocks' dependent load -> store chains run side by side
    const int j = blockIdx.y;
    real_t *d = prm.dst[j] + g0;
    const int64_t sy = tp.zp * tp.tX, sz = tp.tY * tp.tX;
    const real_t *yh_y = tp.yh + (int64_t(tid_y) * nbm + j) * sy;   // (tx, ty-1, tc)
    const real_t *yh_yz = tp.yh + (int64_t(tid_yz) * nbm + j) * sy; // (tx, ty-1, tc-1)
    const real_t *zh_z = tp.zh + (int64_t(tid_z) * nbm + j) * sz;   // (tx, ty, tc-1)
    // the same slabs of the tiles at tx - 1 (one tile earlier in the numbering)
    const int64_t left_y = int64_t(nbm) * sy, left_z = int64_t(nbm) * sz;
    // four rows per thread in flight (the loop is a chain of dependent load -> add -> store otherwise)
    constexpr int U = 4;
    // blockIdx.z splits the two independent parts (y-face rows / z-face plane) over two workgroups
    if (has_y && blockIdx.z == 0) // rows Y = 0, Z >= (has_z ? 1 : 0): contributions of the tiles below in y
      for (int Z0 = rg + has_z; Z0 < Zn; Z0 += U * nrg) {
        real_t s[U], v[U];
        STFEM_UNROLL
        for (int u = 0; u < U; ++u) {
          const int Z = Z0 + u * nrg;
          s[u] = v[u] = real_t(0);
          if (Z < Zn) {
            s[u] = yh_y[Z * tp.tX + X];
            if (dxn) s[u] += (yh_y - left_y)[Z * tp.tX + XpL];
            v[u] = d[plane_stride * Z];
          }
        }
        STFEM_UNROLL
        for (int u = 0; u < U; ++u) {
          const int Z = Z0 + u * nrg;
          if (Z < Zn) d[plane_stride * Z] = v[u] + s[u];
        }
      }
    if (has_z && blockIdx.z == 1) // plane Z = 0: tiles below in z, and for its row Y = 0 also below in y
      for (int Y0 = rg; Y0 < Yn; Y0 += U * nrg) {
        real_t s[U], v[U];
        STFEM_UNROLL
        for (int u = 0; u < U; ++u) {
          const int Y = Y0 + u * nrg;
          s[u] = v[u] = real_t(0);
          if (Y < Yn) {
            s[u] = zh_z[Y * tp.tX + X];
            if (dxn) s[u] += (zh_z - left_z)[Y * tp.tX + XpL];
            if (Y == 0 && has_y) {
              s[u] += yh_y[X] + yh_yz[top_below * tp.tX + X];
              if (dxn) s[u] += (yh_y - left_y)[XpL] + (yh_yz - left_y)[top_below * tp.tX + XpL];
            }
            v[u] = d[int64_t(prm.nx) * Y];
          }
        }
        STFEM_UNROLL
        for (int u = 0; u < U; ++u) {
          const int Y = Y0 + u * nrg;
          if (Y < Yn) d[int64_t(prm.nx) * Y] = v[u] + s[u];
        }
      }
  }
}

template <int P, int NBM, int WV> int launch_tile_w(const SweepParams &prm, const TilePlan &tp0, hipStream_t st)
{
  TilePlan tp = tp0;
  (void)hipGetLastError(); // drop a stale sticky error of an unrelated earlier call
  for (int colour = 1; colour >= 0; --colour) { // odd tiles first: they feed the even ones
    tp.xcolor = colour;
    const int ntxh = (tp.ntx - colour + 1) / 2;
    const int nblocks = ntxh * tp.nty * tp.ntc;
    if (nblocks == 0) continue;
    const bool coef = prm.coef_lap || prm.coef_mass;
#define STFEM_LAUNCH(WW, AA, CC, GG)                                                                        \
  do {                                                                                                     \
    if (colour == 1)                                                                                       \
      hipLaunchKernelGGL((st_sweep_cart_tile<P, NBM, WW, AA, CC, GG, 1>), dim3(nblocks), dim3(TileGeom<P, NBM, GG>::NT), 0, st, prm, tp); \
    else                                                                                                   \
      hipLaunchKernelGGL((st_sweep_cart_tile<P, NBM, WW, AA, CC, GG, 0>), dim3(nblocks), dim3(TileGeom<P, NBM, GG>::NT), 0, st, prm, tp); \
  } while (0)
    if (prm.metric) { // general geometry / per-q coefficients (baked into the metric)
      if (tp.add) STFEM_LAUNCH(STFEM_GEN_WAVES, true, false, true);
      else STFEM_LAUNCH(STFEM_GEN_WAVES, false, false, true);
    } else if (tp.add && coef) STFEM_LAUNCH(WV, true, true, false);
    else if (tp.add) STFEM_LAUNCH(WV, true, false, false);
    else if (coef) STFEM_LAUNCH(WV, false, true, false);
    else STFEM_LAUNCH(WV, false, false, false);
#undef STFEM_LAUNCH
    if (hipGetLastError() != hipSuccess) return -3;
  }
  if (tp.nty > 1 || tp.ntc > 1) {
    hipLaunchKernelGGL((st_tile_fixup<P>), dim3(tp.ntx * tp.nty * tp.ntc, prm.nbo, 2), dim3(256), 0, st, prm, tp, NBM);
    if (hipGetLastError() != hipSuccess) return -3;
  }
  return 0;
}

// Workgroups of the sweep kernel the runtime can keep resident on one CU (registers, LDS): what the
// z-chunk planner fills its rounds with.  STFEM_DEBUG_OCC=1 prints the kernel's resources.
template <int P, int NBM, int WV> int tile_occupancy_w(bool general)
{
  int n = 0;
  hipError_t e;
  const void *kern;
  if (general) {
    auto k = st_sweep_cart_tile<P, NBM, STFEM_GEN_WAVES, false, false, true, 0>;
    kern = reinterpret_cast<const void *>(k);
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, TileGeom<P, NBM, true>::NT, 0);
  } else {
    auto k = st_sweep_cart_tile<P, NBM, WV, false, false, false, 0>;
    kern = reinterpret_cast<const void *>(k);
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, TileGeom<P, NBM, false>::NT, 0);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    n = 0;
  }
  if (getenv("STFEM_DEBUG_OCC")) {
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, kern) == hipSuccess)
      fprintf(stderr, "st_sweep_cart_tile<%d,%d,%s>: %d workgroups/CU; %d registers, %zu B LDS, %zu B scratch\n", P, NBM,
              general ? "general" : "cartesian", n, fa.numRegs, fa.sharedSizeBytes, fa.localSizeBytes);
  }
  return n;
}
template <int P, int NBM> int tile_occupancy_t(bool general)
{
#ifdef STFEM_F32
  return tile_occupancy_w<P, NBM, (P >= 5 ? 2 : 4)>(general); // (FE_Q(5): the 128-register variants spill)
#else
  return tile_occupancy_w<P, NBM, 2>(general); // (the 168-VGPR, three-per-CU variants spill for some (P, NBM))
#endif
}

template <int P, int NBM> int launch_tile_t(const SweepParams &prm, const TilePlan &tp, hipStream_t st)
{
#ifdef STFEM_F32
  // half the registers and half the LDS per workgroup: twice the waves per SIMD
  return launch_tile_w<P, NBM, (P >= 5 ? 2 : 4)>(prm, tp, st);
#else
  return launch_tile_w<P, NBM, 2>(prm, tp, st);
#endif
}

} // namespace

#if !STFEM_TILE_P
namespace {

// One thread per (cell, quadrature point): MappingQ1 Jacobian from the 8 cell vertices, then
// G = c_L w detJ J^-1 J^-T and Mq = c_M w detJ (what FEEvaluation::submit_gradient /
// submit_value multiply with, reference include/operators.h:1149-1163).
__global__ __launch_bounds__(256) void build_metric_kernel(int n, int ncx, int ncy, int64_t ncells,
                                                            const double *__restrict__ vert,
                                                            const double *__restrict__ xq,
                                                            const double *__restrict__ wq,
                                                            const real_t *__restrict__ cl, int cl_layout,
                                                            const real_t *__restrict__ cm, int cm_layout,
                                                            real_t *__restrict__ metric)
{
  const int n3 = n * n * n;
  const int64_t gid = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (gid >= ncells * n3) return;
  const int64_t cell = gid / n3;
  const int q = int(gid - cell * n3);
  const int qx = q % n, qy = (q / n) % n, qz = q / (n * n);
  const int cx = int(cell % ncx), cy = int((cell / ncx) % ncy), cz = int(cell / (int64_t(ncx) * ncy));
  const int64_t nvx = ncx + 1, nvy = ncy + 1;
  const double xi[3] = {xq[qx], xq[qy], xq[qz]};
  double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (int c = 0; c < 2; ++c)
    for (int b = 0; b < 2; ++b)
      for (int a = 0; a < 2; ++a) {
        const double *X = vert + 3 * ((cx + a) + nvx * ((cy + b) + nvy * int64_t(cz + c)));
        const double fx = a ? xi[0] : 1.0 - xi[0], fy = b ? xi[1] : 1.0 - xi[1], fz = c ? xi[2] : 1.0 - xi[2];
        const double dx = a ? 1.0 : -1.0, dy = b ? 1.0 : -1.0, dz = c ? 1.0 : -1.0;
        for (int d = 0; d < 3; ++d) {
          J[d][0] += X[d] * dx * fy * fz;
          J[d][1] += X[d] * fx * dy * fz;
          J[d][2] += X[d] * fx * fy * dz;
        }
      }
  const double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
                     J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
  const double id = 1.0 / det;
  double Ji[3][3]; // Ji[e][d] = d xi_e / d x_d
  Ji[0][0] = (J[1][1] * J[2][2] - J[1][2] * J[2][1]) * id;
  Ji[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id;
  Ji[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
  Ji[1][0] = (J[1][2] * J[2][0] - J[1][0] * J[2][2]) * id;
  Ji[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id;
  Ji[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
  Ji[2][0] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) * id;
  Ji[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id;
  Ji[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
  const double JxW = det * wq[qx] * wq[qy] * wq[qz];
  const double fl = cl_layout == 0 ? 1.0 : (cl_layout == 1 ? cl[cell] : cl[gid]);
  const double fm = cm_layout == 0 ? 1.0 : (cm_layout == 1 ? cm[cell] : cm[gid]);
  real_t *m = metric + (cell * n3 + q) * 8; // one 64-byte record per quadrature point
  int comp = 0;
  for (int e = 0; e < 3; ++e)
    for (int f = e; f < 3; ++f, ++comp)
      m[comp] = real_t(fl * JxW * (Ji[e][0] * Ji[f][0] + Ji[e][1] * Ji[f][1] + Ji[e][2] * Ji[f][2]));
  m[6] = real_t(fm * JxW);
  m[7] = real_t(0);
}

} // namespace

int launch_build_metric(int p, const int nc[3], const double *d_vertices, const double *d_xq,
                        const double *d_wq, const real_t *coef_lap, int lap_layout,
                        const real_t *coef_mass, int mass_layout, real_t *d_metric, void *stream)
{
  const int n = p + 1;
  const int64_t ncells = int64_t(nc[0]) * nc[1] * nc[2], total = ncells * n * n * n;
  (void)hipGetLastError();
  hipLaunchKernelGGL(build_metric_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), n, nc[0], nc[1], ncells, d_vertices, d_xq, d_wq,
                     coef_lap, lap_layout, coef_mass, mass_layout, d_metric);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

int tile_geometry(int p, int nbm, int general, TilePlan &plan)
{
  if (p < 1 || p > 5) return -2;
  nbm = round_nbm(nbm);
  const int n = p + 1;
  const int cb = 64 / n;
  if (nbm > cb) return -2;
  // the store phase maps one slab row to at most 64 lanes
  plan.wx = tile_wx(p, nbm);
  plan.cw = plan.wx * tile_sx(p, nbm, general != 0) * tile_cells_per_wave(p, nbm);
  plan.rows = 4;
  plan.tX = p * plan.cw + 1;
  plan.tY = p * plan.rows + 1;
  return 0;
}

#endif // !STFEM_TILE_P

// The kernel templates are compiled once per degree (make builds this file with
// -DSTFEM_TILE_P=1..5 in parallel); the translation unit without STFEM_TILE_P holds the common
// host code and the dispatcher.
#if STFEM_TILE_P
#define STFEM_PASTE2(a, b) a##b
#define STFEM_PASTE(a, b) STFEM_PASTE2(a, b)
int STFEM_PASTE(launch_cart_tile_p, STFEM_TILE_P)(const SweepParams &prm, const TilePlan &plan, hipStream_t st)
{
  const int nbm = round_nbm(prm.nbi > prm.nbo ? prm.nbi : prm.nbo);
#define STFEM_CASE(NB) \
  if (nbm == NB) return launch_tile_t<STFEM_TILE_P, NB>(prm, plan, st);
#ifdef STFEM_QUICK // development builds: only the two-block instantiation
  STFEM_CASE(2)
#else
  STFEM_CASE(1) STFEM_CASE(2) STFEM_CASE(3) STFEM_CASE(4) STFEM_CASE(6) STFEM_CASE(8)
#endif
#undef STFEM_CASE
  return -2;
}
int STFEM_PASTE(tile_occupancy_p, STFEM_TILE_P)(int nbm_in, int general)
{
  const int nbm = round_nbm(nbm_in);
#define STFEM_CASE(NB) \
  if (nbm == NB) return tile_occupancy_t<STFEM_TILE_P, NB>(general != 0);
#ifdef STFEM_QUICK
  STFEM_CASE(2)
#else
  STFEM_CASE(1) STFEM_CASE(2) STFEM_CASE(3) STFEM_CASE(4) STFEM_CASE(6) STFEM_CASE(8)
#endif
#undef STFEM_CASE
  return 0;
}
#else
int tile_occupancy_p1(int, int);
int tile_occupancy_p2(int, int);
int tile_occupancy_p3(int, int);
int tile_occupancy_p4(int, int);
int tile_occupancy_p5(int, int);
int tile_occupancy(int p, int nbm, int general)
{
  switch (p) {
    case 1: return tile_occupancy_p1(nbm, general);
    case 2: return tile_occupancy_p2(nbm, general);
    case 3: return tile_occupancy_p3(nbm, general);
    case 4: return tile_occupancy_p4(nbm, general);
    case 5: return tile_occupancy_p5(nbm, general);
    default: return 0;
  }
}
int launch_cart_tile_p1(const SweepParams &, const TilePlan &, hipStream_t);
int launch_cart_tile_p2(const SweepParams &, const TilePlan &, hipStream_t);
int launch_cart_tile_p3(const SweepParams &, const TilePlan &, hipStream_t);
int launch_cart_tile_p4(const SweepParams &, const TilePlan &, hipStream_t);
int launch_cart_tile_p5(const SweepParams &, const TilePlan &, hipStream_t);

int launch_cart_tile(int p, const SweepParams &prm, const TilePlan &plan, void *stream)
{
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (p) {
    case 1: return launch_cart_tile_p1(prm, plan, st);
    case 2: return launch_cart_tile_p2(prm, plan, st);
    case 3: return launch_cart_tile_p3(prm, plan, st);
    case 4: return launch_cart_tile_p4(prm, plan, st);
    case 5: return launch_cart_tile_p5(prm, plan, st); // FE_Q(5): this path only (the pencil sweep needs more registers than a wave has)
    default: return -2;
  }
}

const char *cart_tile_name(int, int) { return "st_sweep_cart_tile"; }
#endif

} // namespace STFEM_PREC
} // namespace stfem
