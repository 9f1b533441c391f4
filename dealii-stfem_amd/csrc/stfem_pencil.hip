// "Pencil" variant of the fused space-time cell sweep (Cartesian meshes, cell-wise coefficients),
// the default path since round 2.
//
// Every WAVE owns a pencil of CPW x TY cells in x-y (CPW cells side by side fill the 64 lanes:
// lane = x-node i of cell c of temporal block blk) and marches through the cell layers of a
// z-chunk, TY cells in y per layer.  Nothing is accumulated in LDS and the waves of a workgroup
// never wait for each other inside a layer:
//
//   gather : 8-byte loads, the lanes of a cell row cover P*CPW+1 CONTIGUOUS doubles of a src row
//            (the next cell group is fetched while the current one is in its middle phase);
//   core   : PencilCore (stfem_core.h): fast diagonalisation with two wave-private LDS
//            transposes; the x faces between the cells of the wave are summed inside the slab;
//   y / z  : the y = P row of a cell is the y = 0 row of the next cell of the march, the z = P
//            plane that of the next layer: both stay in the lane's own registers (ycar, zcar);
//   scatter: every finished DoF row leaves straight from registers, once, as a row of
//            contiguous doubles across the lanes.  No atomics, no memset, no LDS slab.
//
// Faces between pencils:
//   y, inside a workgroup (4 waves stacked in y): the top row of wave w goes to an LDS mailbox;
//      wave w+1 keeps its own part of that row back for one layer (4 registers), adds the two
//      after the layer barrier and stores the complete row: one barrier per layer, and only to
//      hand over rows - nobody waits for a phase of another wave;
//   x: every pencil but the first of a row computes the last cell of its left neighbour once more
//      (cell slot 0, the "halo slot": CPW - 1 owned cells per pencil) and so has that cell's
//      contribution to its own x = 0 column without any exchange.  All pencils are independent and
//      run in ONE launch; x-neighbouring pencils have neighbouring block numbers, run at the same
//      time on one XCD and complete each other's cache lines in L2.  (Two colour launches with the
//      faces handed over through slabs - the round-1 scheme - leave every other 200-byte piece of a
//      row for the second launch: measured 1.8 x slower on the memory side alone.)
//   y / z between workgroup tiles: halo slabs (contiguous rows) + st_pencil_fixup, as in the
//      tile variant.
//
// Replaces gather + scatter of MatrixFreeOperator::do_cell_integral_range
// (reference include/operators.h:1112-1133) and the dst = 0 / dst.add(...) traffic of
// SystemMatrix::vmult (operators.h:536-559).
#include "stfem_core.h"

#include <cstdio>
#include <cstdlib>

#ifndef STFEM_PENCIL_P
#define STFEM_PENCIL_P 0
#endif

namespace stfem {
namespace STFEM_PREC {

namespace {

constexpr int PENCIL_WY = 4; // waves of a workgroup, stacked in y

// Diagnostic builds only (tools/build_pencil_exp.sh): -DSTFEM_PENCIL_EXP=<bits> removes parts of the
// kernel for timing (results are wrong), -DSTFEM_PENCIL_TIMELINE records phase timestamps.
//   1 no dst / halo stores   2 no src loads   4 no middle phase   8 no forward / backward phases
//   512 src loads always from the tile's first rows    1024 dst stores always to them
#ifndef STFEM_PENCIL_EXP
#define STFEM_PENCIL_EXP 0
#endif
constexpr int PEX = STFEM_PENCIL_EXP;

// ---- vector memory instructions the compiler does not track ----
// hipcc waits for "all outstanding" (vmcnt(0)) wherever a control-flow path might have issued fewer
// younger operations than another, which here is at every use of a prefetched value: the wave would
// wait for its own just-issued dst stores before it may touch the src planes fetched long before.
// The hot loads and stores are therefore issued from asm statements and waited for with counted
// s_waitcnt vmcnt(N), N = a LOWER bound of the vector-memory instructions issued after the ones
// waited for (vmcnt retires in order, MI355X_MICROARCH.md).  Rules that keep this safe:
//  * a destination register is not mentioned between its load statement and vm_wait + pin
//    (the compiler believes it is defined at the load);
//  * kernels using them must not spill (a spilled destination is stored before the data has
//    landed): csrc/Makefile fails the build otherwise (tools/check_async.py);
//  * only instructions with at least one active lane are counted.
template <typename T> __device__ __forceinline__ void vm_load(T &dst, const T *p)
{
  if constexpr (sizeof(T) == 8) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
  else asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
// lanes outside `mask` (a wave-uniform 64-bit lane mask) keep dst / store nothing
template <typename T> __device__ __forceinline__ void vm_load_masked(T &dst, const T *p, unsigned long long mask)
{
  unsigned long long save;
  if constexpr (sizeof(T) == 8)
    asm volatile("s_mov_b64 %1, exec\n\ts_and_b64 exec, exec, %3\n\tglobal_load_dwordx2 %0, %2, off\n\ts_mov_b64 exec, %1"
                 : "+v"(dst), "=&s"(save) : "v"(p), "s"(mask) : "memory");
  else
    asm volatile("s_mov_b64 %1, exec\n\ts_and_b64 exec, exec, %3\n\tglobal_load_dword %0, %2, off\n\ts_mov_b64 exec, %1"
                 : "+v"(dst), "=&s"(save) : "v"(p), "s"(mask) : "memory");
}
template <typename T> __device__ __forceinline__ void vm_store_masked(T *p, T v, unsigned long long mask)
{
  unsigned long long save;
  if constexpr (sizeof(T) == 8)
    asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %3\n\tglobal_store_dwordx2 %1, %2, off\n\ts_mov_b64 exec, %0"
                 : "=&s"(save) : "v"(p), "v"(v), "s"(mask) : "memory");
  else
    asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %3\n\tglobal_store_dword %1, %2, off\n\ts_mov_b64 exec, %0"
                 : "=&s"(save) : "v"(p), "v"(v), "s"(mask) : "memory");
}
template <int CNT> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory"); }
#ifdef STFEM_PENCIL_TOUCH
// experiment (tools/build_pencil_exp.sh -DSTFEM_PENCIL_TOUCH): a few lanes per row touch the src lines of the cell group AFTER
// the next one (a 4-byte load into a register nobody reads), so that the real loads, one group later, find them in L2
__device__ __forceinline__ void vm_touch(float &sink, const void *p, unsigned long long mask)
{
  unsigned long long save;
  asm volatile("s_mov_b64 %1, exec\n\ts_and_b64 exec, exec, %3\n\tglobal_load_dword %0, %2, off\n\ts_mov_b64 exec, %1"
               : "+v"(sink), "=&s"(save) : "v"(p), "s"(mask) : "memory");
}
#endif

template <int P, int NBM, int TY> struct PencilGeom {
  using G = Geometry<P, NBM>;
  static constexpr int N = P + 1;
  static constexpr int CPW = G::CELLS_PER_WAVE;
  static constexpr int ACTIVE = CPW * NBM * N;
  static constexpr int WY = PENCIL_WY;
  static constexpr int NT = 64 * WY;
  static constexpr int MAIL = 64 * N; // one mailbox buffer: [z][lane]
  // systems of four and more blocks keep the temporal weights in an LDS table [output block][input block][K, M]
  static constexpr bool WLDS = NBM >= 4;
  static constexpr int WTAB = WLDS ? 2 * NBM * NBM : 0;
  // Q2 with three blocks (the velocity components of the Stokes operator): a wave-private tile of the pressure values around the
  // wave's cells of one layer, [2 z-vertices][TY + 1 y-vertices][CPW + 2 x-vertices] (SweepParams::gp)
  static constexpr bool HAS_GRAD = P == 2 && NBM == 3 && sizeof(real_t) == 8;
  static constexpr int GPT_WAVE = HAS_GRAD ? 2 * (TY + 1) * (CPW + 2) : 0;
  // transpose slabs, mailboxes, hand-over counters + tile number, weight table (dynamic LDS: above the 64 KB static limit for Q4)
  static constexpr size_t LDS_BYTES =
    sizeof(real_t) * (size_t(WY) * PencilCore<P, NBM>::LDS_PER_WAVE + size_t(WY - 1) * 2 * MAIL + WTAB + size_t(WY) * GPT_WAVE) +
    sizeof(int) * (2 * WY + 4);
};



// per-lane roles, kept in ONE register (a bool per role would pin two SGPRs each)
enum : unsigned {
  LF_IN = 1,    // feeds an input block
  LF_OUT = 2,   // produces an output block
  LF_LAST = 8,  // last (active) cell of the pencil row: keeps its own x = P slot in the core
  LF_ST = 64,   // stores its column of the finished rows
  LF_XCON = 128 // its column is a Dirichlet column (x faces)
};

template <int P, int NBM, int TY, bool ADD, bool COEF, bool GRAD = false>
__global__ __launch_bounds__(64 * PENCIL_WY, 2)
void st_sweep_pencil(const SweepParams prm, const PencilPlan pp)
{
  using PG = PencilGeom<P, NBM, TY>;
  using Core = PencilCore<P, NBM>;
  constexpr int N = PG::N, NN = N * N, CPW = PG::CPW, WY = PG::WY;
  // LDS: [wave][transpose slab] | [wave 0..WY-2][2 mailbox buffers] | prod[WY], cons[WY], tile number
  extern __shared__ __align__(16) unsigned char smem_raw[];
  real_t *smem = reinterpret_cast<real_t *>(smem_raw);

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  real_t *lds = smem + wave * Core::LDS_PER_WAVE;
  real_t *mail_base = smem + WY * Core::LDS_PER_WAVE;
  real_t *mail_out = mail_base + wave * 2 * PG::MAIL;            // written by waves 0 .. WY-2
  const real_t *mail_in = mail_base + (wave - 1) * 2 * PG::MAIL; // read by waves 1 .. WY-1
  // hand-over counters of the mailboxes: prod[w] = layers whose top row wave w has delivered,
  // cons[w] = layers wave w has taken from the wave below.  No workgroup barrier inside a tile.
  volatile int *flags = reinterpret_cast<volatile int *>(mail_base + (WY - 1) * 2 * PG::MAIL);
  volatile int *prod = flags, *cons = flags + WY;
  volatile int *s_tile = flags + 2 * WY;
  // weight table of the systems with four and more blocks (read-only after the first barrier of the tile loop)
  real_t *wtab = const_cast<real_t *>(reinterpret_cast<volatile real_t *>(flags + 2 * WY + 4));
  [[maybe_unused]] real_t *gpt = wtab + PG::WTAB + wave * PG::GPT_WAVE; // GRAD: this wave's pressure tile
  if constexpr (PG::WLDS) {
    for (int e = threadIdx.x; e < NBM * NBM; e += PG::NT) {
      const int j = e / NBM, q = e % NBM;
      const bool ok = j < prm.nbo && q < prm.nbi;
      wtab[2 * e] = ok ? prm.alpha[j * prm.nbi + q] * prm.vol : real_t(0);
      wtab[2 * e + 1] = ok ? prm.beta[j * prm.nbi + q] * prm.vol : real_t(0);
    }
  }
  auto flag_wait = [&](volatile int *f, int target) {
    while (*f < target) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
  };

  // Workgroup tiles are handed out at run time: the resident workgroups of the launch pull tile
  // numbers until none is left.  Tiles are numbered chunk by chunk (the long z-chunks first, the
  // short ones last, so that the launch ends evenly), x fastest within a chunk.  Every XCD serves one
  // contiguous piece of every chunk with its own counter, so that x-neighbouring pencils run at the
  // same time behind one L2; a workgroup whose XCD has run dry takes tiles of the others.
  const int ncol = pp.ntx * pp.ntyw; // tiles of one z-chunk
  unsigned xcc = 0;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 7;
  for (;;) {
  __syncthreads(); // every wave is done with the previous tile (its mailboxes, the tile number)
  if (threadIdx.x == 0) {
    int t = -1;
    const int per = ncol / 8, rem = ncol % 8;
    for (int q = 0; q < 8 && t < 0; ++q) {
      const int x = (xcc + q) & 7;
      const int lo = x * per + min(x, rem), n = per + (x < rem ? 1 : 0); // this XCD's piece of a chunk
      if (n > 0) {
        const int k = atomicAdd(pp.work + 32 * x, 1);
        if (k < n * pp.ntc) t = (k / n) * ncol + lo + k % n;
      }
    }
    *s_tile = t;
  }
  prod[wave] = 0;
  cons[wave] = 0;
  __syncthreads();
  const int id = __builtin_amdgcn_readfirstlane(*s_tile); // (wave-uniform: keeps everything derived from it in SGPRs)
  if (id < 0) break;
  // pencil tx owns the cells [tx (CPW-1), (tx+1) (CPW-1)) of its row and computes cell tx (CPW-1) - 1 in slot 0 as well
  const int tx = id % pp.ntx, tyw = (id / pp.ntx) % pp.ntyw, tc = id / (pp.ntx * pp.ntyw);
  const int wg_tile = id;
  const int wy = tyw * WY + wave; // global pencil row
  const int cx0 = tx * (CPW - 1) - 1, cy0 = wy * TY;
  const int cz0 = pp.zb[tc];
  const int nlay = pp.zb[tc + 1] - cz0;
  const int nslot = min(CPW, prm.ncx - cx0);          // slots [0, nslot) hold cells (slot 0 of the first pencil does not)
  const int ncy_t = max(0, min(TY, prm.ncy - cy0)); // wave-uniform
  const bool last_x = tx == pp.ntx - 1, last_z = tc == pp.ntc - 1;
  const bool mesh_top = cy0 + TY >= prm.ncy; // this pencil's top row is the mesh's
  // where the top row of this pencil goes: dst (mesh boundary), the next wave's mailbox, the y-halo slab
  const bool top_mail = !mesh_top && wave < WY - 1;
  const bool top_halo = !mesh_top && wave == WY - 1;
  const bool has_lower = wave > 0; // the wave below completes this pencil's y = 0 row one layer late

  const bool lane_ok = lane < PG::ACTIVE;
  const int l = lane_ok ? lane : 0;
  const int i = l % N, c = (l / N) % CPW, blk = l / (N * CPW);
  const int X = P * c + i; // column within the pencil's rows
  unsigned lf = 0;
  {
    const bool cell_ok = lane_ok && c < nslot && cx0 + c >= 0 && ncy_t > 0;
    const int cxl = cx0 + (cell_ok ? c : 1);
    const bool out = cell_ok && blk < prm.nbo;
    // columns of the rows this lane stores: x-node P of a cell is the next cell's node 0; the halo slot
    // stores nothing; the x = P column of the row's last cell exists only at the mesh boundary
    const bool st = out && c > 0 && (i < P || cxl == prm.ncx - 1);
    const bool xcon = ((prm.dmask & 1) && cxl == 0 && i == 0) || ((prm.dmask & 2) && cxl == prm.ncx - 1 && i == P);
    lf = (cell_ok && blk < prm.nbi ? LF_IN : 0) | (out ? LF_OUT : 0) | (c == nslot - 1 ? LF_LAST : 0) |
         (st ? LF_ST : 0) | (xcon ? LF_XCON : 0);
  }
  const int cx = cx0 + ((lf & (LF_IN | LF_OUT)) ? c : 1);

  // GRAD: this lane's part of - gscale B^T p.  x faces are complete when a row leaves the core (the x = P value of a cell was added to
  // the next cell's x = 0 slot in the middle phase), so the lane adds what BOTH x-cells of its node contribute: assembled weights
  // against the pressure vertices p0x .. p0x + 2; in y and z the cell's own contribution, which the carries / mailboxes / halo slabs
  // complete like the rest of the cell's result.
  [[maybe_unused]] real_t gwx[3] = {real_t(0), real_t(0), real_t(0)}, gWy[N][2], gWz[N][2];
  [[maybe_unused]] int gxv = 0;
  if constexpr (GRAD) {
    // (every table entry is picked with selects on compile-time indices: a lane-indexed read of the kernel arguments would be a
    // vector-memory load from the argument segment)
    const int comp = blk < 3 ? blk : 0;
    auto W = [&](int d, int a, int j) { return comp == d ? prm.gw[d][0][a][j] : prm.gw[d][1][a][j]; }; // derivative form along the component's own axis
    const int cxg = cx0 + c;
    const bool lo = cxg > 0;
    const real_t mid0 = W(0, 1, 0), mid1 = W(0, 1, 1);
    const real_t v0 = lo ? W(0, 2, 0) : real_t(0), v1 = (lo ? W(0, 2, 1) : real_t(0)) + W(0, 0, 0), v2 = W(0, 0, 1);
    const real_t e0 = W(0, 2, 0), e1 = W(0, 2, 1); // the x = P column: stored at the mesh boundary only
    gwx[0] = (i == 1 ? mid0 : (i == 0 ? v0 : e0)) * prm.gscale;
    gwx[1] = (i == 1 ? mid1 : (i == 0 ? v1 : e1)) * prm.gscale;
    gwx[2] = (i == 0 ? v2 : real_t(0)) * prm.gscale;
    const int p0x = i == 0 ? cxg - 1 : cxg;
    gxv = min(max(p0x - cx0, 0), CPW - 1); // first of the lane's three x-vertices within the tile (halo-slot lanes: anything in range)
    STFEM_UNROLL
    for (int a = 0; a < N; ++a)
      STFEM_UNROLL
    for (int j = 0; j < 2; ++j) {
      gWy[a][j] = W(1, a, j);
      gWz[a][j] = W(2, a, j);
    }
  }

  // eigenvalue of this lane's z-mode in the middle phase of the core
  real_t lzk = prm.fd_lz[0];
  STFEM_UNROLL
  for (int m = 1; m < N; ++m) lzk = i == m ? prm.fd_lz[m] : lzk;
  // temporal weights of this lane's output block (cell volume folded in); in the LDS table for four and more blocks
  constexpr int NW = PG::WLDS ? 1 : NBM;
  real_t aK0[NW], aM0[NW];
  if constexpr (!PG::WLDS) {
    STFEM_UNROLL
    for (int q = 0; q < NBM; ++q) {
      const bool ok = blk < prm.nbo && q < prm.nbi;
      aK0[q] = ok ? prm.alpha[blk * prm.nbi + q] * prm.vol : real_t(0);
      aM0[q] = ok ? prm.beta[blk * prm.nbi + q] * prm.vol : real_t(0);
    }
  }

  const int64_t plane_stride = int64_t(prm.nx) * prm.ny;
  const int64_t lane_off = int64_t(P) * cx + i + int64_t(prm.nx) * (int64_t(P) * cy0) + plane_stride * (int64_t(P) * cz0);
  const real_t *src_lane = prm.src[(lf & LF_IN) ? blk : 0] + lane_off;
  real_t *dst_lane = prm.dst[(lf & LF_OUT) ? blk : 0] + lane_off;
  const bool xy_boundary = ((prm.dmask & 1) && tx <= 1) || ((prm.dmask & 2) && last_x) ||
                           ((prm.dmask & 4) && cy0 == 0) || ((prm.dmask & 8) && mesh_top);
  const int64_t cells_per_layer = int64_t(prm.ncx) * prm.ncy;
  const int64_t cell0 = cx + int64_t(prm.ncx) * cy0 + cells_per_layer * cz0; // per-cell coefficients

  real_t ycar[N], zcar[TY][P], pend[P], ztop = real_t(0), zfin0 = real_t(0);
  STFEM_UNROLL
  for (int z = 0; z < N; ++z) ycar[z] = real_t(0);
  STFEM_UNROLL
  for (int t = 0; t < TY; ++t)
    STFEM_UNROLL
  for (int y = 0; y < P; ++y) zcar[t][y] = real_t(0);
  STFEM_UNROLL
  for (int z = 0; z < P; ++z) pend[z] = real_t(0);

  // ASYNC: the hot loads / stores are issued from asm statements (see vm_load); the accumulating
  // instantiations (dst += ...) keep compiler-tracked accesses throughout
  constexpr bool ASYNC = !ADD && sizeof(real_t) == 8; // (fp32: tools/check_async.py finds register reuse under in-flight loads)
  constexpr int MAIN_STORES_MIN = P * (P - 1); // dst stores every cell group issues at least
  // One row (fixed y, all z) of a cell group's src planes; `row` walks from row to row.  The pointers
  // are stepped (and made opaque) instead of indexed: with base + uniform offset addressing the
  // compiler keeps a 64-bit offset per row of the gather and of the scatter in SGPRs through the
  // whole loop, far more than there are, and reads them back from VGPR lanes at every use.
  auto step = [&](auto *&q, int64_t n) {
    q += n;
    asm volatile("" : "+v"(q));
  };
  auto load_row = [&](const real_t *&row, int y, real_t (&PA)[NN]) {
    const real_t *q = row;
    STFEM_UNROLL
    for (int z = 0; z < N; ++z) {
      if (PEX & 2) PA[y * N + z] = real_t(y + z);
      else if (ASYNC) vm_load(PA[y * N + z], q);
      else PA[y * N + z] = *q;
      if (z + 1 < N) step(q, plane_stride);
    }
    step(row, prm.nx);
  };
  auto load_coef = [&](int64_t cell, real_t &fK, real_t &fM) { // (operators.h:1060-1087)
    if (ASYNC) {
      if (prm.coef_lap) vm_load(fK, prm.coef_lap + cell);
      if (prm.coef_mass) vm_load(fM, prm.coef_mass + cell);
    } else {
      if (prm.coef_lap) fK = prm.coef_lap[cell];
      if (prm.coef_mass) fM = prm.coef_mass[cell];
    }
  };
  const unsigned long long st_mask = __builtin_amdgcn_ballot_w64((lf & LF_ST) != 0);
#ifdef STFEM_PENCIL_TOUCH
  // one lane per 128-byte line of a block's row: X = 0, 16 and the last column
  const unsigned long long touch_mask = __builtin_amdgcn_ballot_w64((lf & LF_IN) && (X == 0 || X == 16 || X == P * (nslot - 1) + P));
  float touch_sink = 0.0f;
#endif
  real_t sink = real_t(0); // experiments only
  auto put = [&](real_t *q, real_t v) {
    if (PEX & 1) sink += v;
    else if (ADD) *q += v;
    else *q = v;
  };
#ifdef STFEM_PENCIL_TIMELINE
  // [tile][wave][layer][cyl][8] 100 MHz timestamps
#define PTL(k)                                                                                               \
  do {                                                                                                       \
    if (pp.timeline && lane == 0)                                                                            \
      pp.timeline[(((int64_t(wg_tile) * WY + wave) * pp.lz + layer) * TY + cyl) * 8 + (k)] = wall_clock64(); \
  } while (0)
#else
#define PTL(k) do {} while (0)
#endif

  // src planes and coefficients of the current cell group (the next group's are fetched during the
  // middle phase of the current one)
  real_t PA[NN], fK = real_t(1), fM = real_t(1);
  if (ncy_t > 0) {
    const real_t *s = src_lane;
    asm volatile("" : "+v"(s));
    STFEM_UNROLL
    for (int y = 0; y < N; ++y) load_row(s, y, PA);
    if (COEF) load_coef(cell0, fK, fM);
    if (ASYNC) {
      vm_wait<0>();
      pin(PA);
      asm volatile("" : "+v"(fK), "+v"(fM));
    }
  }

  for (int layer = 0; layer < nlay; ++layer) {
    const int cz = cz0 + layer;
    if constexpr (GRAD) { // the pressure values around this wave's cells of the layer (LDS operations of a wave execute in order)
      constexpr int XV = CPW + 2, YV = TY + 1;
      for (int e = lane; e < 2 * YV * XV; e += 64) {
        const int jz = e / (YV * XV), rem = e - jz * (YV * XV), yv = rem / XV, xv = rem - yv * XV;
        const int xg = min(max(cx0 + xv, 0), prm.ncx), yg = min(cy0 + yv, prm.ncy);
        gpt[e] = prm.gp[xg + int64_t(prm.ncx + 1) * (yg + int64_t(prm.ncy + 1) * (cz + jz))];
      }
    }
    const bool last_layer = layer == nlay - 1;
    const bool z_lo = (prm.dmask & 16) && cz == 0, z_hi = (prm.dmask & 32) && cz == prm.ncz - 1;
    const bool masked = xy_boundary || z_lo || z_hi;

    // the row this pencil kept back in the previous layer: the wave below has delivered its part
    if (has_lower && layer > 0 && ncy_t > 0) {
      flag_wait(prod + (wave - 1), layer);
      const real_t *m = mail_in + ((layer - 1) & 1) * PG::MAIL + lane;
      real_t *d = dst_lane + plane_stride * (int64_t(P) * (layer - 1));
      real_t v[P];
      STFEM_UNROLL
      for (int z = 0; z < P; ++z) v[z] = pend[z] + m[64 * z];
      asm volatile("" ::: "memory");
      cons[wave] = layer;
      if (lf & LF_ST) {
        STFEM_UNROLL
        for (int z = 0; z < P; ++z) put(d + plane_stride * z, (lf & LF_XCON) ? real_t(0) : v[z]);
      }
    }

    for (int cyl = 0; cyl < ncy_t; ++cyl) {
      const int cy = cy0 + cyl;
      const bool y_lo = (prm.dmask & 4) && cy == 0, y_hi = (prm.dmask & 8) && cy == prm.ncy - 1;
      // Dirichlet rows / columns of this cell group: constrained DoFs read as 0 and are written as 0
      // (operators.h:1123-1128: read_dof_values / distribute_local_to_global skip them); a face DoF
      // is flagged by both cells that share it, so zeroing the cell results before the carries
      // leaves every carried or handed-over partial sum of a constrained DoF zero too
      auto constrained_dof = [&](int y, int z) -> bool {
        return (lf & LF_XCON) || (y == 0 && y_lo) || (y == P && y_hi) || (z == 0 && z_lo) || (z == P && z_hi);
      };
      PTL(0);
      [[maybe_unused]] real_t ga[2][2]; // GRAD: the x-contracted pressure values of the cell's four (y, z) vertex lines
      if constexpr (GRAD) {
        constexpr int XV = CPW + 2, YV = TY + 1;
        STFEM_UNROLL
        for (int jz = 0; jz < 2; ++jz)
          STFEM_UNROLL
        for (int jy = 0; jy < 2; ++jy) {
          const real_t *q = gpt + (jz * YV + cyl + jy) * XV + gxv;
          ga[jy][jz] = gwx[0] * q[0] + gwx[1] * q[1] + gwx[2] * q[2];
        }
      }
      if (masked) {
        STFEM_UNROLL
        for (int y = 0; y < N; ++y)
          STFEM_UNROLL
        for (int z = 0; z < N; ++z)
          if (constrained_dof(y, z)) PA[y * N + z] = real_t(0);
      }
      real_t aK[NW], aM[NW];
      if constexpr (!PG::WLDS) {
        STFEM_UNROLL
        for (int q = 0; q < NBM; ++q) {
          aK[q] = COEF ? aK0[q] * fK : aK0[q];
          aM[q] = COEF ? aM0[q] * fM : aM0[q];
        }
      }

      real_t *cb_lds = lds + Core::cb_offset(c, blk);
      if (!(PEX & 8)) Core::forward(prm, cb_lds, i, lf & LF_IN, PA);
      else pin(PA);
      PTL(1);

      // PA is free: the next cell group of the march is fetched row by row during the middle phase
      const bool more_y = cyl + 1 < ncy_t;
      const bool has_next = more_y || !last_layer;
      const int64_t next_rows = more_y ? int64_t(P) * (cyl + 1) : 0, next_planes = more_y ? int64_t(P) * layer : int64_t(P) * (layer + 1);
      const real_t *sn = (PEX & 512) ? src_lane : src_lane + plane_stride * next_planes + int64_t(prm.nx) * next_rows; // walks
      asm volatile("" : "+v"(sn));
      auto prefetch_row = [&](int y) {
        if (!has_next) return;
        load_row(sn, y, PA);
        if (COEF && y == N - 1)
          load_coef(cell0 + (more_y ? int64_t(prm.ncx) * (cyl + 1) + cells_per_layer * layer : cells_per_layer * (layer + 1)), fK, fM);
      };
      PTL(2);
      if constexpr (PG::WLDS) {
        // (fK, fM are the CURRENT cell's: the hook overwrites them with the next cell's during the phase)
        const real_t cK = COEF ? fK : real_t(1), cM = COEF ? fM : real_t(1);
        Core::middle_stream(prm, lds, c, blk, i, lf & LF_OUT, lf & LF_LAST, lzk, wtab + blk * (2 * NBM), cK, cM, prefetch_row);
      } else if (!(PEX & 4)) Core::middle(prm, lds, c, blk, i, lf & LF_OUT, lf & LF_LAST, lzk, aK, aM, prefetch_row);
      else {
        STFEM_UNROLL
        for (int y = 0; y < N; ++y) prefetch_row(y);
      }
      PTL(3);

#ifdef STFEM_PENCIL_TOUCH
      // the group after the next one
      const int g2 = cyl + 2, l2 = layer + g2 / ncy_t, c2 = g2 % ncy_t;
      const bool has_next2 = l2 < nlay;
      const real_t *tn = src_lane + plane_stride * (int64_t(P) * l2) + int64_t(prm.nx) * (int64_t(P) * c2);
      asm volatile("" : "+v"(tn));
      auto touch_row = [&](int y) {
        if (!has_next2) return;
        const real_t *q = tn;
        STFEM_UNROLL
        for (int z = 0; z < N; ++z) {
          vm_touch(touch_sink, q, touch_mask);
          if (z + 1 < N) step(q, plane_stride);
        }
        step(tn, prm.nx);
      };
#endif
      // finished rows leave as they come out of the last sweep: y, z in [0, P).  The y = 0 row of a
      // pencil with a wave below is kept back for one layer.
      const bool defer = has_lower && cyl == 0;
      real_t *d = (PEX & 1024) ? dst_lane // (1024: every group overwrites the same rows: the lines stay in L2)
                               : dst_lane + plane_stride * (int64_t(P) * layer) + int64_t(prm.nx) * (int64_t(P) * cyl);
      asm volatile("" : "+v"(d));
      real_t *drow = d; // walks from row to row
      real_t znew[P];
      auto row_done = [&](int y, real_t (&r)[N]) {
#ifdef STFEM_PENCIL_TOUCH
        touch_row(y);
#endif
        if constexpr (GRAD) { // this cell's gradient term at the row's nodes (before the constrained DoFs are zeroed)
          const real_t b0 = gWy[y][0] * ga[0][0] + gWy[y][1] * ga[1][0], b1 = gWy[y][0] * ga[0][1] + gWy[y][1] * ga[1][1];
          STFEM_UNROLL
          for (int z = 0; z < N; ++z) r[z] += gWz[z][0] * b0 + gWz[z][1] * b1;
        }
        if (masked) {
          STFEM_UNROLL
          for (int z = 0; z < N; ++z)
            if (constrained_dof(y, z)) r[z] = real_t(0);
        }
        // faces shared with the previous cell of the march (y) and with the previous layer (z)
        if (y == 0 && cyl > 0) {
          STFEM_UNROLL
          for (int z = 0; z < N; ++z) r[z] += ycar[z];
        }
        if (y == P) {
          STFEM_UNROLL
          for (int z = 0; z < N; ++z) ycar[z] = r[z];
          return;
        }
        if (layer > 0) r[0] += zcar[0][y];
        znew[y] = r[P];
        if (y == 0 && defer) {
          STFEM_UNROLL
          for (int z = 0; z < P; ++z) pend[z] = r[z];
          zfin0 = r[P];
          step(drow, prm.nx);
          return;
        }
        real_t *q = drow;
        STFEM_UNROLL
        for (int z = 0; z < P; ++z) {
          if (ASYNC && !(PEX & 1)) vm_store_masked(q, r[z], st_mask);
          else if (lf & LF_ST) put(q, r[z]);
          step(q, plane_stride);
        }
        // top plane of the chunk: to dst on the last chunk, to the z-halo slab otherwise
        if (last_layer && (lf & LF_ST)) {
          if (last_z) put(q, r[P]);
          else if (!(PEX & 1))
            (pp.zh + ((int64_t(wg_tile) * NBM + blk) * pp.tYW + P * (TY * wave + cyl) + y) * pp.tX + X)[0] = r[P];
        }
        step(drow, prm.nx);
      };
      if (!(PEX & 8)) Core::backward(prm, cb_lds, i, row_done);
      else {
        STFEM_UNROLL
        for (int y = 0; y < N; ++y) {
          real_t r[N];
          STFEM_UNROLL
          for (int z = 0; z < N; ++z) r[z] = lzk + real_t(y + z);
          pin(r);
          row_done(y, r);
        }
      }
      PTL(4);
      STFEM_UNROLL
      for (int t = 0; t + 1 < TY; ++t)
        STFEM_UNROLL
      for (int y = 0; y < P; ++y) zcar[t][y] = zcar[t + 1][y];
      STFEM_UNROLL
      for (int y = 0; y < P; ++y) zcar[TY - 1][y] = znew[y];

      // the prefetched src planes are older than this group's dst stores
      if (ASYNC && !(PEX & 2)) {
        if (PEX & 1) vm_wait<0>();
#ifdef STFEM_PENCIL_TOUCH
        else if (has_next2) vm_wait<MAIN_STORES_MIN + NN>(); // the touches are younger than the prefetched planes too
#endif
        else vm_wait<MAIN_STORES_MIN>();
        pin(PA);
        asm volatile("" : "+v"(fK), "+v"(fM));
      }
      PTL(5);
      // top row of the pencil
      if (cyl == ncy_t - 1) {
        real_t yrow[N];
        STFEM_UNROLL
        for (int z = 0; z < N; ++z) yrow[z] = ycar[z];
        if (layer > 0) yrow[0] += ztop;
        ztop = yrow[P];
        if (top_mail) {
          if (layer >= 2) flag_wait(cons + (wave + 1), layer - 1); // the buffer's previous content has been taken
          real_t *m = mail_out + (layer & 1) * PG::MAIL + lane;
          STFEM_UNROLL
          for (int z = 0; z < N; ++z) m[64 * z] = yrow[z];
          asm volatile("" ::: "memory");
          prod[wave] = layer + 1;
        } else if (lf & LF_ST) {
          if (top_halo) {
            real_t *yh = pp.yh + ((int64_t(wg_tile) * NBM + blk) * pp.zp + P * layer) * pp.tX + X;
            STFEM_UNROLL
            for (int z = 0; z < N; ++z)
              if ((z < P || last_layer) && !(PEX & 1)) yh[z * pp.tX] = yrow[z];
          } else { // the mesh's top row
            STFEM_UNROLL
            for (int z = 0; z < P; ++z) put(d + plane_stride * z + int64_t(prm.nx) * P, yrow[z]);
            if (last_layer) {
              if (last_z) put(d + plane_stride * P + int64_t(prm.nx) * P, yrow[P]);
              else (pp.zh + ((int64_t(wg_tile) * NBM + blk) * pp.tYW + P * (TY * wave + cyl) + P) * pp.tX + X)[0] = yrow[P];
            }
          }
        }
      }
    }
    // pencils with fewer than TY cell rows keep the z-carry slots aligned
    for (int r = ncy_t; r < TY && ncy_t > 0; ++r) {
      real_t t0[P];
      STFEM_UNROLL
      for (int y = 0; y < P; ++y) t0[y] = zcar[0][y];
      STFEM_UNROLL
      for (int t = 0; t + 1 < TY; ++t)
        STFEM_UNROLL
      for (int y = 0; y < P; ++y) zcar[t][y] = zcar[t + 1][y];
      STFEM_UNROLL
      for (int y = 0; y < P; ++y) zcar[TY - 1][y] = t0[y];
    }
  }
  if (PEX && sink == real_t(1.2345e30)) pp.zh[0] = sink; // experiment sink, never true
#ifdef STFEM_PENCIL_TOUCH
  asm volatile("" ::"v"(touch_sink)); // (the register stays reserved while touches can be in flight)
#endif

  // last layer's kept-back row, and the y = 0 row of the chunk's top plane
  if (has_lower && ncy_t > 0) {
    flag_wait(prod + (wave - 1), nlay);
    const real_t *m = mail_in + ((nlay - 1) & 1) * PG::MAIL + lane;
    real_t *d = dst_lane + plane_stride * (int64_t(P) * (nlay - 1));
    real_t v[N];
    STFEM_UNROLL
    for (int z = 0; z < P; ++z) v[z] = pend[z] + m[64 * z];
    v[P] = zfin0 + m[64 * P];
    if (lf & LF_ST) {
      const bool xc = lf & LF_XCON;
      STFEM_UNROLL
      for (int z = 0; z < P; ++z) put(d + plane_stride * z, xc ? real_t(0) : v[z]);
      if (last_z) put(d + plane_stride * P, xc ? real_t(0) : v[P]);
      else (pp.zh + ((int64_t(wg_tile) * NBM + blk) * pp.tYW + P * TY * wave) * pp.tX + X)[0] = xc ? real_t(0) : v[P];
    }
  }
  } // next tile
}

// Adds the halo partial sums of the workgroup tiles below in y / z to the rows a tile owns on its
// y = 0 and z = 0 faces (contiguous in x).  One workgroup per (tile, block, face part).
template <int P>
__global__ __launch_bounds__(256) void st_pencil_fixup(const SweepParams prm, const PencilPlan pp, int nbm, int cpw, int ty)
{
  const int id = blockIdx.x;
  // the sweep that fed this launch is complete: zero its tile counters for the next sweep
  if (id == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x < 8) pp.work[32 * threadIdx.x] = 0;
  const int tx = id % pp.ntx, tyw = (id / pp.ntx) % pp.ntyw, tc = id / (pp.ntx * pp.ntyw);
  const int has_y = tyw > 0, has_z = tc > 0;
  if (!(has_y | has_z)) return;
  const int cyw = ty * PENCIL_WY; // cell rows of a workgroup tile
  const int cx0 = tx * (cpw - 1) - 1, cy0 = tyw * cyw, cz0 = pp.zb[tc];
  const int nslot = min(cpw, prm.ncx - cx0), ncy_t = min(cyw, prm.ncy - cy0);
  const int nlay = pp.zb[tc + 1] - cz0;
  const bool last_x = tx == pp.ntx - 1, last_y = tyw == pp.ntyw - 1, last_z = tc == pp.ntc - 1;
  // stored columns of the tile's rows (see LF_ST in the sweep): everything but the halo slot, and the
  // mesh's last column
  const int x_begin = P, x_end = P * nslot + (last_x ? 1 : 0);
  const int Yn = P * ncy_t + (last_y ? 1 : 0), Zn = P * nlay + (last_z ? 1 : 0);
  const int lpr = pp.tX <= 16 ? 16 : (pp.tX <= 32 ? 32 : 64), nrg = 256 / lpr; // lanes per row: narrow tiles (Q3 x 3 blocks: 13 columns) keep more rows per workgroup
  const int X = threadIdx.x % lpr, rg = threadIdx.x / lpr;
  if (X < x_begin || X >= x_end) return;
  const int64_t plane_stride = int64_t(prm.nx) * prm.ny;
  const int64_t g0 = int64_t(P) * cx0 + X + int64_t(prm.nx) * (int64_t(P) * cy0) + plane_stride * (int64_t(P) * cz0);
  const int tid_y = id - pp.ntx, tid_z = id - pp.ntx * pp.ntyw, tid_yz = tid_z - pp.ntx;
  const int top_below = has_z ? P * (cz0 - pp.zb[tc - 1]) : 0;
  const int j = blockIdx.y;
  real_t *d = prm.dst[j] + g0;
  const int64_t sy = int64_t(pp.zp) * pp.tX, sz = int64_t(pp.tYW) * pp.tX;
  const real_t *yh_y = pp.yh + (int64_t(tid_y) * nbm + j) * sy + X;   // (tx, tyw-1, tc)
  const real_t *yh_yz = pp.yh + (int64_t(tid_yz) * nbm + j) * sy + X; // (tx, tyw-1, tc-1)
  const real_t *zh_z = pp.zh + (int64_t(tid_z) * nbm + j) * sz + X;   // (tx, tyw, tc-1)
  constexpr int U = 4;
  if (has_y && blockIdx.z == 0) // rows Y = 0, Z >= (has_z ? 1 : 0)
    for (int Z0 = rg + has_z; Z0 < Zn; Z0 += U * nrg) {
      real_t s[U], v[U];
      STFEM_UNROLL
      for (int u = 0; u < U; ++u) {
        const int Z = Z0 + u * nrg;
        s[u] = v[u] = real_t(0);
        if (Z < Zn) {
          s[u] = yh_y[Z * pp.tX];
          v[u] = d[plane_stride * Z];
        }
      }
      STFEM_UNROLL
      for (int u = 0; u < U; ++u) {
        const int Z = Z0 + u * nrg;
        if (Z < Zn) d[plane_stride * Z] = v[u] + s[u];
      }
    }
  if (has_z && blockIdx.z == 1) // plane Z = 0
    for (int Y0 = rg; Y0 < Yn; Y0 += U * nrg) {
      real_t s[U], v[U];
      STFEM_UNROLL
      for (int u = 0; u < U; ++u) {
        const int Y = Y0 + u * nrg;
        s[u] = v[u] = real_t(0);
        if (Y < Yn) {
          s[u] = zh_z[Y * pp.tX];
          if (Y == 0 && has_y) s[u] += yh_y[0] + yh_yz[top_below * pp.tX];
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

template <int P, int NBM, int TY> int launch_pencil_ty(const SweepParams &prm, const PencilPlan &pp, hipStream_t st)
{
  using PG = PencilGeom<P, NBM, TY>;
  (void)hipGetLastError();
  const bool coef = prm.coef_lap || prm.coef_mass;
  const int nblocks = pp.ntx * pp.ntyw * pp.ntc;
  const int grid = nblocks < pp.grid ? nblocks : pp.grid;
#define STFEM_LAUNCH(AA, CC)                                                                                       \
  do {                                                                                                            \
    static bool lds_set = false;                                                                                  \
    if (!lds_set) {                                                                                               \
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(&st_sweep_pencil<P, NBM, TY, AA, CC>),               \
                              hipFuncAttributeMaxDynamicSharedMemorySize, int(PG::LDS_BYTES)) != hipSuccess)      \
        return -3;                                                                                                \
      lds_set = true;                                                                                             \
    }                                                                                                             \
    hipLaunchKernelGGL((st_sweep_pencil<P, NBM, TY, AA, CC>), dim3(grid), dim3(PG::NT), PG::LDS_BYTES, st, prm, pp); \
  } while (0)
  if constexpr (PG::HAS_GRAD) {
    if (prm.gp) { // (the caller sets gp only for dst = ..., no coefficients: csrc/stfem_capi.hip)
      if (pp.add || coef) return -2;
      static bool lds_set = false;
      if (!lds_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&st_sweep_pencil<P, NBM, TY, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                int(PG::LDS_BYTES)) != hipSuccess)
          return -3;
        lds_set = true;
      }
      hipLaunchKernelGGL((st_sweep_pencil<P, NBM, TY, false, false, true>), dim3(grid), dim3(PG::NT), PG::LDS_BYTES, st, prm, pp);
    }
  }
  if (PG::HAS_GRAD && prm.gp) {
  } else if (pp.add && coef) STFEM_LAUNCH(true, true);
  else if (pp.add) STFEM_LAUNCH(true, false);
  else if (coef) STFEM_LAUNCH(false, true);
  else STFEM_LAUNCH(false, false);
#undef STFEM_LAUNCH
  // The tile counters are zero on entry: zeroed at allocation and again after every sweep - by the fix-up kernel, or
  // by a memset where there is no fix-up or a launch has failed (counters left at their end values would make every
  // later sweep on this context pull no tile and return with dst unwritten).
  const bool sweep_ok = hipGetLastError() == hipSuccess;
  bool fixed_up = false;
  if (sweep_ok && (pp.ntyw > 1 || pp.ntc > 1)) {
    hipLaunchKernelGGL((st_pencil_fixup<P>), dim3(nblocks, prm.nbo, 2), dim3(256), 0, st, prm, pp, NBM, PG::CPW, TY);
    fixed_up = hipGetLastError() == hipSuccess;
    if (!fixed_up) (void)hipMemsetAsync(pp.work, 0, 8 * 32 * sizeof(int), st);
    return fixed_up ? 0 : -3;
  }
  if (hipMemsetAsync(pp.work, 0, 8 * 32 * sizeof(int), st) != hipSuccess || !sweep_ok) return -3;
  return 0;
}

// cell rows per pencil: two, except where the second set of z-carry registers makes the kernel spill
// (Q4 with three temporal blocks; tools/check_async.py fails the build if an instantiation spills)
// Q1 / Q2 pencils of up to four blocks take four cell rows: fewer mailbox rows and y-halo rows per cell where the work per cell group
// is small (measured, Q2 on 144^3 cells: cG(1) x 4 steps 1.27 -> 1.16 ms, cG(2) 0.471 -> 0.443 ms; profiles/r3/experiments.txt).
// pencil_ty: the largest instantiated value = the default; STFEM_PENCIL_TY selects a smaller instantiated one (1, 2, 4).
constexpr int pencil_ty(int p, int nbm) { return (p == 4 && nbm == 3) ? 1 : ((p <= 2 && nbm <= 4) ? 4 : 2); }
constexpr int pencil_ty_default(int p, int nbm) { return pencil_ty(p, nbm); }

template <int P, int NBM> int launch_pencil_t(const SweepParams &prm, const PencilPlan &pp, hipStream_t st)
{
  if constexpr (Geometry<P, NBM>::CELLS_PER_WAVE < 2) return -2; // needs the halo slot and at least one owned cell
  else {
    if (pp.ty == 1) return launch_pencil_ty<P, NBM, 1>(prm, pp, st);
    if constexpr (pencil_ty(P, NBM) >= 2)
      if (pp.ty == 2) return launch_pencil_ty<P, NBM, 2>(prm, pp, st);
    if constexpr (pencil_ty(P, NBM) >= 4)
      if (pp.ty == 4) return launch_pencil_ty<P, NBM, 4>(prm, pp, st);
    return -2;
  }
}

} // namespace

#if STFEM_PENCIL_P
#define STFEM_PASTE2(a, b) a##b
#define STFEM_PASTE(a, b) STFEM_PASTE2(a, b)
int STFEM_PASTE(launch_pencil_p, STFEM_PENCIL_P)(const SweepParams &prm, const PencilPlan &plan, hipStream_t st)
{
  const int nbm = round_nbm(prm.nbi > prm.nbo ? prm.nbi : prm.nbo);
#define STFEM_CASE(NB) \
  if (nbm == NB) return launch_pencil_t<STFEM_PENCIL_P, NB>(prm, plan, st);
#ifdef STFEM_QUICK
  STFEM_CASE(2)
#else
  STFEM_CASE(1) STFEM_CASE(2) STFEM_CASE(3) STFEM_CASE(4) STFEM_CASE(6) STFEM_CASE(8)
#endif
#undef STFEM_CASE
  return -2;
}
#else
int launch_pencil_p1(const SweepParams &, const PencilPlan &, hipStream_t);
int launch_pencil_p2(const SweepParams &, const PencilPlan &, hipStream_t);
int launch_pencil_p3(const SweepParams &, const PencilPlan &, hipStream_t);
int launch_pencil_p4(const SweepParams &, const PencilPlan &, hipStream_t);

int launch_pencil(int p, const SweepParams &prm, const PencilPlan &plan, void *stream)
{
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (p) {
    case 1: return launch_pencil_p1(prm, plan, st);
    case 2: return launch_pencil_p2(prm, plan, st);
    case 3: return launch_pencil_p3(prm, plan, st);
    case 4: return launch_pencil_p4(prm, plan, st);
    default: return -2;
  }
}

// cells per wave in x (0: no pencil instantiation for this (p, nbm)); fills the slab extents
int pencil_geometry(int p, int nbm, int ty, PencilPlan &plan)
{
  if (p < 1 || p > 4) return -2;
  nbm = round_nbm(nbm);
  if (ty < 1 || ty > pencil_ty(p, nbm) || ty == 3) ty = pencil_ty_default(p, nbm);
  // (four and more temporal blocks per launch: PencilCore::middle_stream; the caller cuts systems for which
  // fewer than two cells fit a wave - Q4 with seven or eight blocks - into smaller panels)
  const int cpw = (64 / (p + 1)) / nbm;
  if (cpw < 2) return -2;
  plan.cpw = cpw;
  plan.ty = ty;
  plan.tX = p * cpw + 1;
  plan.tYW = p * ty * PENCIL_WY + 1;
  return 0;
}
#endif

} // namespace STFEM_PREC
} // namespace stfem
