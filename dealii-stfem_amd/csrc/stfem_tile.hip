// "Tile" variant of the fused space-time cell sweep (Cartesian meshes), the default path.
//
// A 256-thread workgroup (4 waves) owns a tile of CW x 4 cells in x-y (one wave per cell row)
// and marches through LZ cell layers in z:
//
//   per layer:  gather src planes from HBM (coalesced along x, faces shared through L1/L2)
//               -> cell_core (registers + wave-private LDS transposes)
//               -> accumulate the (p+1)^3 results of all cells of the layer in an LDS slab
//                  (owner lane writes, the other sharers ds_add_f64)
//               -> stream the four finished DoF planes of the layer to HBM, rows of
//                  p*CW+1 contiguous doubles; the fifth plane is carried to the next layer.
//
// DoFs on the tile's upper x / y / z face belong to the neighbouring tile: their partial sums are
// written (plain stores) to per-tile halo slabs and added to the owner's value by
// st_tile_fixup.  Every DoF of dst is therefore written exactly once by plain stores: no global
// atomics, no memset, bit-reproducible results up to the order of the LDS adds.
//
// Replaces the scatter of MatrixFreeOperator::do_cell_integral_range
// (reference include/operators.h:1112-1133, distribute_local_to_global) and the dst = 0 /
// dst.add(...) traffic of SystemMatrix::vmult (operators.h:536-559).
#include "stfem_core.h"

namespace stfem {

namespace {

template <int P, int NBM> struct TileGeom {
  using G = Geometry<P, NBM>;
  static constexpr int N = P + 1;
  static constexpr int CW = G::CELLS_PER_WAVE;
  static constexpr int ROWS = G::WAVES;
  static constexpr int TX = P * CW + 1;
  static constexpr int TY = P * ROWS + 1;
  static constexpr int PLANE = TX * TY;
  static constexpr int ACC = NBM * N * PLANE;                 // accumulation slab (aliases trans)
  static constexpr int TRANS = G::WAVES * G::LDS_PER_WAVE;    // transpose slabs
  static constexpr int MAIN = ACC > TRANS ? ACC : TRANS;
  static constexpr int CARRY = NBM * PLANE;
  static constexpr int LDS_DOUBLES = MAIN + CARRY;
};

struct TileCoords {
  int tx, ty, tc;          // tile indices
  int cx0, cy0, cz0;       // first cell
  int ncx, ncy, nlay;      // active cells / layers in this tile
  bool last_x, last_y, last_z;
};

__device__ __forceinline__ TileCoords tile_coords(const SweepParams &prm, const TilePlan &tp, int id)
{
  TileCoords t;
  t.tx = id % tp.ntx;
  t.ty = (id / tp.ntx) % tp.nty;
  t.tc = id / (tp.ntx * tp.nty);
  t.cx0 = t.tx * tp.cw;
  t.cy0 = t.ty * tp.rows;
  t.cz0 = t.tc * tp.lz;
  t.ncx = min(tp.cw, prm.ncx - t.cx0);
  t.ncy = min(tp.rows, prm.ncy - t.cy0);
  t.nlay = min(tp.lz, prm.ncz - t.cz0);
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
  // XCDs 0..rem-1 hold per+1 tiles
  const int start = xcd * per + min(xcd, rem);
  return start + slot;
}

template <int P, int NBM>
__global__ __launch_bounds__(256, 2) void st_sweep_cart_tile(const SweepParams prm, const TilePlan tp)
{
  using TG = TileGeom<P, NBM>;
  using G = Geometry<P, NBM>;
  constexpr int N = TG::N;
  constexpr int TX = TG::TX, TY = TG::TY, PLANE = TG::PLANE;
  __shared__ double smem[TG::LDS_DOUBLES];
  double *acc = smem;              // [blk][k][Y][X], aliases the transpose slabs
  double *carry = smem + TG::MAIN; // [blk][Y][X]: top plane of the previous layer

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  double *lds = smem + wave * G::LDS_PER_WAVE;

  const int nblocks = tp.ntx * tp.nty * tp.ntc;
  const TileCoords t = tile_coords(prm, tp, logical_block(blockIdx.x, nblocks));

  const bool lane_ok = lane < G::ACTIVE;
  const int l = lane_ok ? lane : 0;
  const int k = l % N;
  const int blk = (l / N) % NBM;
  const int cxl = l / (N * NBM); // cell within the wave's row
  const int cyl = wave;
  const bool cell_ok = lane_ok && cxl < t.ncx && cyl < t.ncy;
  const int cx = t.cx0 + (cell_ok ? cxl : 0), cy = t.cy0 + (cell_ok ? cyl : 0);
  const bool in_active = cell_ok && blk < prm.nbi;
  const bool out_active = cell_ok && blk < prm.nbo;
  const int64_t cell_xy = cx + int64_t(prm.ncx) * cy;
  const int64_t cells_per_layer = int64_t(prm.ncx) * prm.ncy;

  double aK0[NBM], aM0[NBM];
  STFEM_UNROLL
  for (int i = 0; i < NBM; ++i) {
    const bool ok = blk < prm.nbo && i < prm.nbi;
    aK0[i] = ok ? prm.alpha[blk * prm.nbi + i] * prm.vol : 0.0;
    aM0[i] = ok ? prm.beta[blk * prm.nbi + i] * prm.vol : 0.0;
  }

  // which entries of this lane's result plane it initialises in the LDS slab ("owner")
  const bool own_x_hi = cxl == t.ncx - 1; // last active cell of the row owns its x = P column
  const bool own_y_hi = cyl == t.ncy - 1;

  for (int e = tid; e < TG::CARRY; e += 256) carry[e] = 0.0;
  __syncthreads();

  const int64_t plane_stride = int64_t(prm.nx) * prm.ny;
  const int64_t xy_base = int64_t(P) * cx + int64_t(prm.nx) * (int64_t(P) * cy);
  const double *src_blk = prm.src[in_active ? blk : 0];

  for (int layer = 0; layer < t.nlay; ++layer) {
    const int cz = t.cz0 + layer;
    PlaneMask pm = plane_mask<P>(prm, cx, cy, cz, k);
    double PA[N * N];
    {
      const double *s = src_blk + xy_base + plane_stride * (int64_t(P) * cz + k);
      STFEM_UNROLL
      for (int y = 0; y < N; ++y)
        STFEM_UNROLL
      for (int x = 0; x < N; ++x) {
        const double v = in_active ? s[int64_t(y) * prm.nx + x] : 0.0;
        PA[y * N + x] = constrained<P>(pm, y, x) ? 0.0 : v;
      }
    }
    double aK[NBM], aM[NBM];
    {
      const int64_t c = cell_xy + cells_per_layer * cz;
      const double fK = prm.coef_lap ? prm.coef_lap[c] : 1.0;
      const double fM = prm.coef_mass ? prm.coef_mass[c] : 1.0;
      STFEM_UNROLL
      for (int i = 0; i < NBM; ++i) {
        aK[i] = aK0[i] * fK;
        aM[i] = aM0[i] * fM;
      }
    }

    cell_core<P, NBM>(prm, lds, cxl, blk, k, in_active, out_active, aK, aM, PA);

    __syncthreads(); // all waves are done with the transpose slabs: the region becomes `acc`

    // owner lanes initialise their DoFs (adding the plane carried from the previous layer)
    double *a = acc + ((blk * N + k) * TY + P * cyl) * TX + P * cxl;
    const double *cr = carry + (blk * TY + P * cyl) * TX + P * cxl;
    if (out_active) {
      STFEM_UNROLL
      for (int y = 0; y < N; ++y)
        STFEM_UNROLL
      for (int x = 0; x < N; ++x) {
        const bool owned = (x < P || own_x_hi) && (y < P || own_y_hi);
        if (owned) {
          double v = constrained<P>(pm, y, x) ? 0.0 : PA[y * N + x];
          if (k == 0) v += cr[y * TX + x];
          a[y * TX + x] = v;
        }
      }
    }
    __syncthreads();
    // the other sharers of a face / edge / vertex DoF add their part
    if (out_active) {
      STFEM_UNROLL
      for (int y = 0; y < N; ++y)
        STFEM_UNROLL
      for (int x = 0; x < N; ++x) {
        const bool owned = (x < P || own_x_hi) && (y < P || own_y_hi);
        if ((x == P || y == P) && !owned && !constrained<P>(pm, y, x))
          atomicAdd(&a[y * TX + x], PA[y * N + x]); // ds_add_f64
      }
    }
    __syncthreads();

    // stream the finished planes k = 0..P-1 (and k = P on the last layer) to their destination
    const bool last_layer = layer == t.nlay - 1;
    const int xext = P * t.ncx, yext = P * t.ncy; // highest local index in use
    for (int e = tid; e < NBM * N * PLANE; e += 256) {
      const int X = e % TX, r = e / TX;
      const int Y = r % TY, kk = (r / TY) % N, j = r / (TY * N);
      if (j >= prm.nbo || X > xext || Y > yext) continue;
      const double v = acc[e];
      if (kk == P && !last_layer) {
        carry[(j * TY + Y) * TX + X] = v;
        continue;
      }
      const int zl = P * layer + kk; // chunk-local plane
      const int tile_id = t.tx + tp.ntx * (t.ty + tp.nty * t.tc);
      if (X == xext && !t.last_x) {
        tp.xh[((int64_t(tile_id) * NBM + j) * tp.zp + zl) * tp.tY + Y] = v;
      } else if (Y == yext && !t.last_y) {
        tp.yh[((int64_t(tile_id) * NBM + j) * tp.zp + zl) * tp.tX + X] = v;
      } else if (kk == P && !t.last_z) { // only on the last layer
        tp.zh[((int64_t(tile_id) * NBM + j) * tp.tY + Y) * tp.tX + X] = v;
      } else {
        const int64_t g = int64_t(P) * t.cx0 + X + int64_t(prm.nx) * (int64_t(P) * t.cy0 + Y) +
                          plane_stride * (int64_t(P) * t.cz0 + zl);
        double *d = prm.dst[j] + g;
        *d = tp.add ? *d + v : v;
      }
    }
    __syncthreads(); // slab free again for the next layer's transposes
  }
}

// Adds the halo partial sums of the lower neighbours to the DoFs a tile owns on its x = 0,
// y = 0 and z = 0 faces.  One workgroup per tile.
template <int P>
__global__ __launch_bounds__(256) void st_tile_fixup(const SweepParams prm, const TilePlan tp, int nbm)
{
  const TileCoords t = tile_coords(prm, tp, blockIdx.x);
  const int has_x = t.tx > 0, has_y = t.ty > 0, has_z = t.tc > 0;
  if (!(has_x | has_y | has_z)) return;
  // owned local extents
  const int Xn = P * t.ncx + (t.last_x ? 1 : 0), Yn = P * t.ncy + (t.last_y ? 1 : 0),
            Zn = P * t.nlay + (t.last_z ? 1 : 0);
  const int xs = has_x ? 1 : 0, ys = has_y ? 1 : 0;
  const int nfx = has_x ? Yn * Zn : 0;                      // X = 0, all Y, Z
  const int nfy = has_y ? (Xn - xs) * Zn : 0;               // Y = 0, X >= xs
  const int nfz = has_z ? (Xn - xs) * (Yn - ys) : 0;        // Z = 0, X >= xs, Y >= ys
  const int64_t plane_stride = int64_t(prm.nx) * prm.ny;
  for (int idx = threadIdx.x; idx < nfx + nfy + nfz; idx += blockDim.x) {
    int X, Y, Z;
    if (idx < nfx) {
      X = 0; Y = idx % Yn; Z = idx / Yn;
    } else if (idx < nfx + nfy) {
      const int q = idx - nfx;
      Y = 0; X = xs + q % (Xn - xs); Z = q / (Xn - xs);
    } else {
      const int q = idx - nfx - nfy;
      Z = 0; X = xs + q % (Xn - xs); Y = ys + q / (Xn - xs);
    }
    const int64_t g = int64_t(P) * t.cx0 + X + int64_t(prm.nx) * (int64_t(P) * t.cy0 + Y) +
                      plane_stride * (int64_t(P) * t.cz0 + Z);
    for (int j = 0; j < prm.nbo; ++j) {
      double s = 0.0;
      for (int dz = 0; dz <= (Z == 0 ? has_z : 0); ++dz)
        for (int dy = 0; dy <= (Y == 0 ? has_y : 0); ++dy)
          for (int dx = 0; dx <= (X == 0 ? has_x : 0); ++dx) {
            if (!(dx | dy | dz)) continue;
            const int ntx_ = t.tx - dx, nty_ = t.ty - dy, ntc_ = t.tc - dz;
            const int nid = ntx_ + tp.ntx * (nty_ + tp.nty * ntc_);
            // the DoF in the neighbour's local coordinates (neighbours below are never ragged)
            const int Xp = dx ? P * tp.cw : X, Yp = dy ? P * tp.rows : Y, Zp = dz ? P * tp.lz : Z;
            const int64_t base = int64_t(nid) * nbm + j;
            if (dx) s += tp.xh[(base * tp.zp + Zp) * tp.tY + Yp];
            else if (dy) s += tp.yh[(base * tp.zp + Zp) * tp.tX + Xp];
            else s += tp.zh[(base * tp.tY + Yp) * tp.tX + Xp];
          }
      prm.dst[j][g] += s;
    }
  }
}

template <int P, int NBM> int launch_tile_t(const SweepParams &prm, const TilePlan &tp, hipStream_t st)
{
  const int nblocks = tp.ntx * tp.nty * tp.ntc;
  hipLaunchKernelGGL((st_sweep_cart_tile<P, NBM>), dim3(nblocks), dim3(256), 0, st, prm, tp);
  if (hipGetLastError() != hipSuccess) return -3;
  if (tp.ntx > 1 || tp.nty > 1 || tp.ntc > 1) {
    hipLaunchKernelGGL((st_tile_fixup<P>), dim3(nblocks), dim3(256), 0, st, prm, tp, NBM);
    if (hipGetLastError() != hipSuccess) return -3;
  }
  return 0;
}

} // namespace

int tile_geometry(int p, int nbm, TilePlan &plan)
{
  if (p < 1 || p > 4) return -2;
  nbm = round_nbm(nbm);
  const int n = p + 1;
  const int cb = 64 / n;
  if (nbm > cb) return -2;
  plan.cw = cb / nbm;
  plan.rows = 4;
  plan.tX = p * plan.cw + 1;
  plan.tY = p * plan.rows + 1;
  return 0;
}

int launch_cart_tile(int p, const SweepParams &prm, const TilePlan &plan, void *stream)
{
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nbm = round_nbm(prm.nbi > prm.nbo ? prm.nbi : prm.nbo);
#define STFEM_CASE(PP, NB) \
  if (p == PP && nbm == NB) return launch_tile_t<PP, NB>(prm, plan, st);
#define STFEM_CASES(PP) \
  STFEM_CASE(PP, 1) STFEM_CASE(PP, 2) STFEM_CASE(PP, 3) STFEM_CASE(PP, 4) STFEM_CASE(PP, 6) STFEM_CASE(PP, 8)
  STFEM_CASES(1)
  STFEM_CASES(2)
  STFEM_CASES(3)
  STFEM_CASES(4)
#undef STFEM_CASES
#undef STFEM_CASE
  return -2;
}

const char *cart_tile_name(int, int) { return "st_sweep_cart_tile"; }

} // namespace stfem
