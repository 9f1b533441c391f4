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
//   x: pencils are 2-coloured by the parity of their x index; odd pencils run first and leave
//      the partial sums of their two end faces in x-slabs, in the (y-mode, z-mode) form they have
//      in the middle of the core (N*N values per cell face); even pencils add them there and
//      store complete rows;
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
//   16 no layer barrier      32 no x-slab traffic
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

template <int P, int NBM, int TY> struct PencilGeom {
  using G = Geometry<P, NBM>;
  static constexpr int N = P + 1;
  static constexpr int CPW = G::CELLS_PER_WAVE;
  static constexpr int ACTIVE = CPW * NBM * N;
  static constexpr int WY = PENCIL_WY;
  static constexpr int NT = 64 * WY;
  static constexpr int LDS_PER_WAVE = CPW * NBM * G::CBS; // transpose slab of one wave
  static constexpr int MAIL = 64 * N;                     // one mailbox buffer: [z][lane]
  static constexpr int LDS_DOUBLES = WY * LDS_PER_WAVE + (WY - 1) * 2 * MAIL;
};

__device__ __forceinline__ int pchunk_begin(int c, int ncz, int ntc) { return int(int64_t(c) * ncz / ntc); }

// blocks b, b+8, ... share an XCD: give every XCD one contiguous range of workgroup tiles
__device__ __forceinline__ int plogical_block(int b, int nblocks)
{
  const int per = nblocks / 8, rem = nblocks % 8;
  const int xcd = b % 8, slot = b / 8;
  return xcd * per + min(xcd, rem) + slot;
}

// only LDS traffic has to be complete at the layer barrier: global loads and stores stay in flight
__device__ __forceinline__ void lds_barrier()
{
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// per-lane roles, kept in ONE register (a bool per role would pin two SGPRs each)
enum : unsigned {
  LF_IN = 1,      // feeds an input block
  LF_OUT = 2,     // produces an output block
  LF_FIRST = 4,   // first cell of the pencil row
  LF_LAST = 8,    // last (active) cell of the pencil row
  LF_ADDLO = 16,  // even pencil: receives the left neighbour's face
  LF_ADDHI = 32,  // even pencil: receives the right neighbour's face
  LF_ST = 64,     // stores its column of the finished rows
  LF_XCON = 128,  // its column is a Dirichlet column (x faces)
  LF_XS = 256     // reads (even) / writes (odd) an x-slab
};

template <int P, int NBM, int TY, bool ADD, bool COEF, int COLOR>
__global__ __launch_bounds__(64 * PENCIL_WY, 2)
void st_sweep_pencil(const SweepParams prm, const PencilPlan pp)
{
  using PG = PencilGeom<P, NBM, TY>;
  using Core = PencilCore<P, NBM>;
  constexpr int N = PG::N, NN = N * N, CPW = PG::CPW, WY = PG::WY;
  constexpr bool odd = COLOR == 1;
  __shared__ real_t smem[PG::LDS_DOUBLES];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  real_t *lds = smem + wave * PG::LDS_PER_WAVE;
  real_t *mail_base = smem + WY * PG::LDS_PER_WAVE;
  real_t *mail_out = mail_base + wave * 2 * PG::MAIL;            // written by waves 0 .. WY-2
  const real_t *mail_in = mail_base + (wave - 1) * 2 * PG::MAIL; // read by waves 1 .. WY-1

  // workgroup tile of this launch's x colour
  // (experiment 64: all pencils in ONE launch, x faces wrong - what would a colour-free decomposition cost?)
  const int ntxh = (PEX & 64) ? pp.ntx : (pp.ntx - COLOR + 1) / 2;
  const int nblocks = ntxh * pp.ntyw * pp.ntc;
  const int id = plogical_block(blockIdx.x, nblocks);
  const int tx = (PEX & 64) ? id % ntxh : 2 * (id % ntxh) + COLOR, tyw = (id / ntxh) % pp.ntyw, tc = id / (ntxh * pp.ntyw);
  const int wg_tile = tx + pp.ntx * (tyw + pp.ntyw * tc);
  const int wy = tyw * WY + wave; // global pencil row
  const int cx0 = tx * CPW, cy0 = wy * TY;
  const int cz0 = pchunk_begin(tc, prm.ncz, pp.ntc);
  const int nlay = pchunk_begin(tc + 1, prm.ncz, pp.ntc) - cz0;
  const int ncx_t = min(CPW, prm.ncx - cx0);
  const int ncy_t = max(0, min(TY, prm.ncy - cy0)); // wave-uniform
  const bool last_x = tx == pp.ntx - 1, last_z = tc == pp.ntc - 1;
  const bool mesh_top = cy0 + TY >= prm.ncy; // this pencil's top row is the mesh's
  // where the top row of this pencil goes: dst (mesh boundary), the next wave's mailbox, the y-halo slab
  const bool top_mail = !mesh_top && wave < WY - 1;
  const bool top_halo = !mesh_top && wave == WY - 1;
  const bool has_lower = wave > 0; // the wave below completes this pencil's y = 0 row one layer late
  const bool has_left = tx > 0, has_right = !last_x;

  const bool lane_ok = lane < PG::ACTIVE;
  const int l = lane_ok ? lane : 0;
  const int i = l % N, c = (l / N) % CPW, blk = l / (N * CPW);
  const int X = P * c + i; // column within the pencil's rows
  unsigned lf = 0;
  {
    const bool cell_ok = lane_ok && c < ncx_t && ncy_t > 0;
    const int cxl = cx0 + (cell_ok ? c : 0);
    const bool out = cell_ok && blk < prm.nbo;
    const bool first = c == 0, last = c == ncx_t - 1;
    // columns of the rows this lane stores: x-node P of a cell is the next cell's node 0; the two end
    // columns of the pencil belong to the even pencils (or to the mesh boundary)
    const bool st = out && ((i < P && !(X == 0 && odd && has_left)) || (i == P && last && (!odd || last_x)));
    const bool xcon = ((prm.dmask & 1) && cxl == 0 && i == 0) || ((prm.dmask & 2) && cxl == prm.ncx - 1 && i == P);
    const bool lo = first && has_left, hi = last && has_right;
    lf = (cell_ok && blk < prm.nbi ? LF_IN : 0) | (out ? LF_OUT : 0) | (first ? LF_FIRST : 0) | (last ? LF_LAST : 0) |
         (!odd && lo ? LF_ADDLO : 0) | (!odd && hi ? LF_ADDHI : 0) | (st ? LF_ST : 0) | (xcon ? LF_XCON : 0) |
         (out && (lo || hi) ? LF_XS : 0);
  }
  const int cx = cx0 + ((lf & (LF_IN | LF_OUT)) ? c : 0);

  // eigenvalue of this lane's z-mode in the middle phase of the core
  real_t lzk = prm.fd_lz[0];
  STFEM_UNROLL
  for (int m = 1; m < N; ++m) lzk = i == m ? prm.fd_lz[m] : lzk;
  // temporal weights of this lane's output block (cell volume folded in)
  real_t aK0[NBM], aM0[NBM];
  STFEM_UNROLL
  for (int q = 0; q < NBM; ++q) {
    const bool ok = blk < prm.nbo && q < prm.nbi;
    aK0[q] = ok ? prm.alpha[blk * prm.nbi + q] * prm.vol : real_t(0);
    aM0[q] = ok ? prm.beta[blk * prm.nbi + q] * prm.vol : real_t(0);
  }

  const int64_t plane_stride = int64_t(prm.nx) * prm.ny;
  const int64_t lane_off = int64_t(P) * cx + i + int64_t(prm.nx) * (int64_t(P) * cy0) + plane_stride * (int64_t(P) * cz0);
  const real_t *src_lane = prm.src[(lf & LF_IN) ? blk : 0] + lane_off;
  real_t *dst_lane = prm.dst[(lf & LF_OUT) ? blk : 0] + lane_off;
  const bool xy_boundary = ((prm.dmask & 1) && tx == 0) || ((prm.dmask & 2) && last_x) ||
                           ((prm.dmask & 4) && cy0 == 0) || ((prm.dmask & 8) && mesh_top);
  const int64_t cells_per_layer = int64_t(prm.ncx) * prm.ncy;

  // x-slabs: [chunk][pencil row][tx][blk][layer][cyl][k][XS] (N values, padded to an even count).
  // Even pencils read the odd neighbours' (first cell: left neighbour's right face, last cell: right
  // neighbour's left face); odd pencils write their own.
  constexpr int XS = N + (N & 1);
  real_t *xs_ptr = nullptr;
  {
    const int64_t xs_tile = int64_t(NBM) * pp.lz * TY * N * XS;
    const bool lo = lf & LF_FIRST; // (a one-cell pencil is always the last of its row: its left face)
    real_t *base = odd ? (lo && has_left ? pp.xl : pp.xr) : (lo && has_left ? pp.xr : pp.xl);
    const int tx_ = odd ? tx : (lo && has_left ? tx - 1 : tx + 1);
    if (lf & LF_XS)
      xs_ptr = base + ((int64_t(tc) * (pp.ntyw * WY) + wy) * pp.ntx + tx_) * xs_tile + (int64_t(blk) * pp.lz * TY * N + i) * XS;
  }

  real_t ycar[N], zcar[TY][P], pend[P], ztop = real_t(0), zfin0 = real_t(0);
  STFEM_UNROLL
  for (int z = 0; z < N; ++z) ycar[z] = real_t(0);
  STFEM_UNROLL
  for (int t = 0; t < TY; ++t)
    STFEM_UNROLL
  for (int y = 0; y < P; ++y) zcar[t][y] = real_t(0);
  STFEM_UNROLL
  for (int z = 0; z < P; ++z) pend[z] = real_t(0);

  // (the group's base pointers are made opaque: loop strength reduction otherwise keeps one 64-bit
  // induction pointer per row of the gather and of the scatter alive through the whole loop, ~80 VGPRs)
  // ASYNC: the hot loads / stores are issued from asm statements (see vm_load); the accumulating
  // instantiations (dst += ...) keep compiler-tracked accesses throughout
  constexpr bool ASYNC = !ADD;
  constexpr int MAIN_STORES_MIN = P * (P - 1); // dst stores every cell group issues at least
  auto load_group = [&](const real_t *s, real_t (&PA)[NN]) {
    asm volatile("" : "+v"(s));
    STFEM_UNROLL
    for (int y = 0; y < N; ++y)
      STFEM_UNROLL
    for (int z = 0; z < N; ++z) {
      if (PEX & 2) PA[y * N + z] = real_t(y + z);
      else if (ASYNC) vm_load(PA[y * N + z], s + (plane_stride * z + int64_t(prm.nx) * y));
      else PA[y * N + z] = s[plane_stride * z + int64_t(prm.nx) * y];
    }
  };
  const unsigned long long st_mask = __builtin_amdgcn_ballot_w64((lf & LF_ST) != 0);
  const unsigned long long xs_mask = __builtin_amdgcn_ballot_w64((lf & LF_XS) != 0);
  real_t sink = real_t(0); // experiments only
  auto put = [&](real_t *q, real_t v) {
    if (PEX & 1) sink += v;
    else if (ADD) *q += v;
    else *q = v;
  };
#ifdef STFEM_PENCIL_TIMELINE
  // [block][wave][layer][cyl][8] 100 MHz timestamps of the even-colour launch
#define PTL(k)                                                                                          \
  do {                                                                                                  \
    if (COLOR == 0 && pp.timeline && lane == 0)                                                         \
      pp.timeline[(((int64_t(blockIdx.x) * WY + wave) * pp.lz + layer) * TY + cyl) * 8 + (k)] = wall_clock64(); \
  } while (0)
#else
#define PTL(k) do {} while (0)
#endif

  real_t PA[NN];
  if (ncy_t > 0) {
    load_group(src_lane, PA);
    if (ASYNC) {
      vm_wait<0>();
      pin(PA);
    }
  }

  for (int layer = 0; layer < nlay; ++layer) {
    const int cz = cz0 + layer;
    const bool last_layer = layer == nlay - 1;
    const bool z_lo = (prm.dmask & 16) && cz == 0, z_hi = (prm.dmask & 32) && cz == prm.ncz - 1;
    const bool masked = xy_boundary || z_lo || z_hi;

    // the row this pencil kept back in the previous layer: the wave below has delivered its part
    if (has_lower && layer > 0 && ncy_t > 0) {
      const real_t *m = mail_in + ((layer - 1) & 1) * PG::MAIL + lane;
      real_t *d = dst_lane + plane_stride * (int64_t(P) * (layer - 1));
      real_t v[P];
      STFEM_UNROLL
      for (int z = 0; z < P; ++z) v[z] = pend[z] + m[64 * z];
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
      auto zero_constrained = [&](real_t (&A)[NN]) {
        const bool xc = lf & LF_XCON;
        STFEM_UNROLL
        for (int y = 0; y < N; ++y)
          STFEM_UNROLL
        for (int z = 0; z < N; ++z)
          if (xc || (y == 0 && y_lo) || (y == P && y_hi) || (z == 0 && z_lo) || (z == P && z_hi)) A[y * N + z] = real_t(0);
      };
      PTL(0);
      if (masked) zero_constrained(PA);

      // even pencils: the odd neighbours' face sums of this cell group (issued before the src
      // prefetch below, so that waiting for them does not wait for the prefetch)
      real_t xin[N], xout[N];
      STFEM_UNROLL
      for (int y = 0; y < N; ++y) xin[y] = xout[y] = real_t(0);
      const int xs_group = (layer * TY + cyl) * N * XS;
      if (!odd && !(PEX & 32)) {
        if (ASYNC) {
          STFEM_UNROLL
          for (int y = 0; y < N; ++y) vm_load_masked(xin[y], xs_ptr + (xs_group + y), xs_mask);
        } else if (lf & LF_XS) {
          STFEM_UNROLL
          for (int y = 0; y < N; ++y) xin[y] = xs_ptr[xs_group + y];
        }
      }

      // per-cell coefficients (operators.h:1060-1087), folded into the temporal weights in the middle
      // phase; fetched like the slab values (older than the prefetch, waited for together with them)
      real_t fK = real_t(1), fM = real_t(1);
      if (COEF) {
        const int64_t cell = cx + int64_t(prm.ncx) * cy + cells_per_layer * cz;
        if (ASYNC) {
          if (prm.coef_lap) vm_load(fK, prm.coef_lap + cell);
          if (prm.coef_mass) vm_load(fM, prm.coef_mass + cell);
        } else {
          if (prm.coef_lap) fK = prm.coef_lap[cell];
          if (prm.coef_mass) fM = prm.coef_mass[cell];
        }
      }

      real_t *cb_lds = lds + (c * NBM + blk) * Core::CBS;
      if (!(PEX & 8)) Core::forward(prm, cb_lds, i, lf & LF_IN, PA);
      else pin(PA);
      PTL(1);

      // PA is free: fetch the next cell group of the march (lands during the middle phase)
      {
        const bool more_y = cyl + 1 < ncy_t;
        if (more_y || !last_layer) {
          const real_t *s = more_y ? src_lane + plane_stride * (int64_t(P) * layer) + int64_t(prm.nx) * (int64_t(P) * (cyl + 1))
                                   : src_lane + plane_stride * (int64_t(P) * (layer + 1));
          load_group(s, PA);
        }
        // the slab values are older than the prefetch: they have landed when at most the prefetch is in flight
        if (ASYNC && ((!odd && !(PEX & 32)) || COEF)) {
          if (more_y || !last_layer) vm_wait<NN>();
          else vm_wait<0>();
          pin(xin);
          asm volatile("" : "+v"(fK), "+v"(fM));
        }
      }
      real_t aK[NBM], aM[NBM];
      STFEM_UNROLL
      for (int q = 0; q < NBM; ++q) {
        aK[q] = COEF ? aK0[q] * fK : aK0[q];
        aM[q] = COEF ? aM0[q] * fM : aM0[q];
      }

      PTL(2);
      if (!(PEX & 4))
        Core::template middle<!odd, odd>(prm, lds, c, blk, i, lf & LF_OUT, lf & LF_FIRST, lf & LF_LAST, lf & LF_ADDLO,
                                         lf & LF_ADDHI, lzk, aK, aM, xin, xout);
      PTL(3);

      if (odd && !(PEX & 32)) {
        if (ASYNC) {
          STFEM_UNROLL
          for (int y = 0; y < N; ++y) vm_store_masked(xs_ptr + (xs_group + y), xout[y], xs_mask);
        } else if (lf & LF_XS) {
          STFEM_UNROLL
          for (int y = 0; y < N; ++y) xs_ptr[xs_group + y] = xout[y];
        }
      }

      real_t R[NN];
      if (!(PEX & 8)) Core::backward(prm, cb_lds, i, R);
      else {
        STFEM_UNROLL
        for (int e = 0; e < NN; ++e) R[e] = xin[e % N] + real_t(e);
        pin(R);
      }
      if (masked) zero_constrained(R);
      PTL(4);

      // faces shared with the previous cell of the march (y) and with the previous layer (z)
      if (cyl > 0) {
        STFEM_UNROLL
        for (int z = 0; z < N; ++z) R[0 * N + z] += ycar[z];
      }
      if (layer > 0) {
        STFEM_UNROLL
        for (int y = 0; y < P; ++y) R[y * N + 0] += zcar[0][y];
      }
      STFEM_UNROLL
      for (int z = 0; z < N; ++z) ycar[z] = R[P * N + z];
      STFEM_UNROLL
      for (int t = 0; t + 1 < TY; ++t)
        STFEM_UNROLL
      for (int y = 0; y < P; ++y) zcar[t][y] = zcar[t + 1][y];
      STFEM_UNROLL
      for (int y = 0; y < P; ++y) zcar[TY - 1][y] = R[y * N + P];

      // finished rows: y, z in [0, P).  The y = 0 row of a pencil with a wave below is kept back.
      const bool defer = has_lower && cyl == 0;
      real_t *d = dst_lane + plane_stride * (int64_t(P) * layer) + int64_t(prm.nx) * (int64_t(P) * cyl);
      asm volatile("" : "+v"(d));
      if (defer) {
        STFEM_UNROLL
        for (int z = 0; z < P; ++z) pend[z] = R[0 * N + z];
        zfin0 = R[0 * N + P];
      }
      if (ASYNC && !(PEX & 1)) {
        STFEM_UNROLL
        for (int z = 0; z < P; ++z)
          STFEM_UNROLL
        for (int y = 0; y < P; ++y) {
          if (y == 0 && defer) continue;
          vm_store_masked(d + (plane_stride * z + int64_t(prm.nx) * y), R[y * N + z], st_mask);
        }
      }
      if (lf & LF_ST) {
        if (!ASYNC || (PEX & 1)) {
          STFEM_UNROLL
          for (int z = 0; z < P; ++z)
            STFEM_UNROLL
          for (int y = 0; y < P; ++y) {
            if (y == 0 && defer) continue;
            put(d + plane_stride * z + int64_t(prm.nx) * y, R[y * N + z]);
          }
        }
        // top plane of the chunk: to dst on the last chunk, to the z-halo slab otherwise
        if (last_layer) {
          real_t *zh = pp.zh + ((int64_t(wg_tile) * NBM + blk) * pp.tYW + P * (TY * wave + cyl)) * pp.tX + X;
          STFEM_UNROLL
          for (int y = 0; y < P; ++y) {
            if (y == 0 && defer) continue;
            if (last_z) put(d + plane_stride * P + int64_t(prm.nx) * y, R[y * N + P]);
            else if (!(PEX & 1)) zh[y * pp.tX] = R[y * N + P];
          }
        }
      }
      // the prefetched src planes are older than this group's dst stores
      if (ASYNC && !(PEX & 2)) {
        if (PEX & 1) vm_wait<0>();
        else vm_wait<MAIN_STORES_MIN>();
        pin(PA);
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
          real_t *m = mail_out + (layer & 1) * PG::MAIL + lane;
          STFEM_UNROLL
          for (int z = 0; z < N; ++z) m[64 * z] = yrow[z];
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
#ifdef STFEM_PENCIL_TIMELINE
    {
      const int cyl = 0;
      PTL(6);
    }
#endif
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
    if (!(PEX & 16)) lds_barrier(); // the mailboxes of this layer are complete
#ifdef STFEM_PENCIL_TIMELINE
    {
      const int cyl = 0;
      PTL(7);
    }
#endif
  }
  if (PEX && sink == real_t(1.2345e30)) pp.zh[0] = sink; // experiment sink, never true

  // last layer's kept-back row, and the y = 0 row of the chunk's top plane
  if (has_lower && ncy_t > 0) {
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
}

// Adds the halo partial sums of the workgroup tiles below in y / z to the rows a tile owns on its
// y = 0 and z = 0 faces (contiguous in x).  One workgroup per (tile, block, face part).
template <int P>
__global__ __launch_bounds__(256) void st_pencil_fixup(const SweepParams prm, const PencilPlan pp, int nbm, int cpw, int ty)
{
  const int id = blockIdx.x;
  const int tx = id % pp.ntx, tyw = (id / pp.ntx) % pp.ntyw, tc = id / (pp.ntx * pp.ntyw);
  const int has_y = tyw > 0, has_z = tc > 0;
  if (!(has_y | has_z)) return;
  const int cyw = ty * PENCIL_WY; // cell rows of a workgroup tile
  const int cx0 = tx * cpw, cy0 = tyw * cyw, cz0 = pchunk_begin(tc, prm.ncz, pp.ntc);
  const int ncx_t = min(cpw, prm.ncx - cx0), ncy_t = min(cyw, prm.ncy - cy0);
  const int nlay = pchunk_begin(tc + 1, prm.ncz, pp.ntc) - cz0;
  const bool last_x = tx == pp.ntx - 1, last_y = tyw == pp.ntyw - 1, last_z = tc == pp.ntc - 1;
  const bool odd = tx & 1;
  // stored columns of the tile's rows (see st_lane in the sweep)
  const int x_begin = (odd && tx > 0) ? 1 : 0, x_end = P * ncx_t + ((!odd || last_x) ? 1 : 0);
  const int Yn = P * ncy_t + (last_y ? 1 : 0), Zn = P * nlay + (last_z ? 1 : 0);
  const int lpr = pp.tX <= 32 ? 32 : 64, nrg = 256 / lpr;
  const int X = threadIdx.x % lpr, rg = threadIdx.x / lpr;
  if (X < x_begin || X >= x_end) return;
  const int64_t plane_stride = int64_t(prm.nx) * prm.ny;
  const int64_t g0 = int64_t(P) * cx0 + X + int64_t(prm.nx) * (int64_t(P) * cy0) + plane_stride * (int64_t(P) * cz0);
  const int tid_y = id - pp.ntx, tid_z = id - pp.ntx * pp.ntyw, tid_yz = tid_z - pp.ntx;
  const int top_below = has_z ? P * (cz0 - pchunk_begin(tc - 1, prm.ncz, pp.ntc)) : 0;
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

template <int P, int NBM, int TY> int launch_pencil_ty(const SweepParams &prm, const PencilPlan &pp0, hipStream_t st)
{
  using PG = PencilGeom<P, NBM, TY>;
  PencilPlan pp = pp0;
  (void)hipGetLastError();
  const bool coef = prm.coef_lap || prm.coef_mass;
  for (int colour = 1; colour >= 0; --colour) { // odd pencils first: they feed the even ones
    const int ntxh = (PEX & 64) ? (colour == 0 ? pp.ntx : 0) : (pp.ntx - colour + 1) / 2;
    const int nblocks = ntxh * pp.ntyw * pp.ntc;
    if (nblocks == 0) continue;
#define STFEM_LAUNCH(AA, CC)                                                                                   \
  do {                                                                                                        \
    if (colour == 1)                                                                                          \
      hipLaunchKernelGGL((st_sweep_pencil<P, NBM, TY, AA, CC, 1>), dim3(nblocks), dim3(PG::NT), 0, st, prm, pp); \
    else                                                                                                      \
      hipLaunchKernelGGL((st_sweep_pencil<P, NBM, TY, AA, CC, 0>), dim3(nblocks), dim3(PG::NT), 0, st, prm, pp); \
  } while (0)
    if (pp.add && coef) STFEM_LAUNCH(true, true);
    else if (pp.add) STFEM_LAUNCH(true, false);
    else if (coef) STFEM_LAUNCH(false, true);
    else STFEM_LAUNCH(false, false);
#undef STFEM_LAUNCH
    if (hipGetLastError() != hipSuccess) return -3;
  }
  if (pp.ntyw > 1 || pp.ntc > 1) {
    hipLaunchKernelGGL((st_pencil_fixup<P>), dim3(pp.ntx * pp.ntyw * pp.ntc, prm.nbo, 2), dim3(256), 0, st, prm, pp, NBM,
                       PG::CPW, TY);
    if (hipGetLastError() != hipSuccess) return -3;
  }
  return 0;
}

template <int P, int NBM> int launch_pencil_t(const SweepParams &prm, const PencilPlan &pp, hipStream_t st)
{
  if (Geometry<P, NBM>::CELLS_PER_WAVE < 2) return -2; // one cell per wave: both end faces in one lane (tile variant)
  switch (pp.ty) {
    case 1: return launch_pencil_ty<P, NBM, 1>(prm, pp, st);
    case 2: return launch_pencil_ty<P, NBM, 2>(prm, pp, st);
    default: return -2;
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
  STFEM_CASE(1) STFEM_CASE(2) STFEM_CASE(3)
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
  // more than three temporal blocks per launch: the middle phase of the core (one x-line of every
  // input block in flight per lane) no longer fits 256 VGPRs without spilling; the tile variant
  // handles those systems
  if (nbm > 3) return -2;
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
