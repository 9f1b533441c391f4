// Cross-lane contraction of 5-lane groups in fp64 on MI355X (gfx950): wave shuffles against the
// wave-private LDS transpose the sweep kernels use (BASELINE north_star names
// "wavefront shuffles for the 1D contractions").
//
// The operation: every lane i of a group of N = 5 lanes (one Q4 cell line along x) holds a 5 x 5 plane
// x[y][z] of doubles; wanted is  out_i[y][z] = sum_j A[i][j] x_j[y][z]  for a dense 5 x 5 matrix A
// (the x transform of the fast diagonalisation; 25 values per lane).
//
//  lds     : write the plane to LDS (25 ds_write_b64), wave fence, every lane reads the x-lines of its own
//            y-row back (25 ds_read_b64: lane i takes row y = i) and contracts them in registers with the
//            even-odd kernel (17 fp64 instructions per line) - what PencilCore::middle does;
//  bpermute: every lane fetches the four other lanes' values with ds_bpermute_b32 (2 per double) and
//            does 5 FMAs with its own row of A (lane-dependent coefficients in VGPRs);
//  dpp     : the groups are packed three per 16-lane DPP row (lane 15 idle); shifted copies of the wave's
//            register by s = -4 .. 4 (row_shr / row_shl, 2 v_mov_dpp per double) and 9 FMAs with the
//            lane's coefficient A[i][i+s] (0 outside the group).
//
// Output: ns per plane contraction and wave (all 4 waves of a workgroup busy, 2 workgroups per CU as in the
// sweep), and the instruction counts per plane.  hipcc --offload-arch=gfx950 -O3 -o shuffle_bench shuffle_bench.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

constexpr int N = 5, NN = 25;

__device__ __forceinline__ void wave_fence()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// even-odd contraction of one line with a persymmetric 5 x 5 matrix (13 constants), 17 instructions
__device__ __forceinline__ void eo5(const double (&c)[13], const double (&x)[N], double (&y)[N])
{
  const double e0 = x[0] + x[4], e1 = x[1] + x[3], o0 = x[0] - x[4], o1 = x[1] - x[3];
  const double a0 = fma(c[2], x[2], fma(c[1], e1, c[0] * e0)), a1 = fma(c[5], x[2], fma(c[4], e1, c[3] * e0));
  const double b0 = fma(c[7], o1, c[6] * o0), b1 = fma(c[9], o1, c[8] * o0);
  y[0] = a0 + b0; y[4] = a0 - b0; y[1] = a1 + b1; y[3] = a1 - b1;
  y[2] = fma(c[12], x[2], fma(c[11], e1, c[10] * e0));
}

__global__ __launch_bounds__(256, 2) void k_lds(double *out, int iters)
{
  __shared__ double slab[4][12 * 165]; // 12 groups per wave, padded planes (PencilCore: PS = 33)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / N, i = lane % N;
  double *cb = slab[wave] + (g < 12 ? g : 0) * 165;
  double c[13];
  for (int q = 0; q < 13; ++q) c[q] = 0.1 + 0.01 * q;
  double P[NN];
  for (int e = 0; e < NN; ++e) P[e] = 1.0 + 1e-3 * e + 1e-6 * threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    if (lane < 60) {
#pragma unroll
      for (int y = 0; y < N; ++y)
#pragma unroll
        for (int z = 0; z < N; ++z) cb[i * 33 + y * N + z] = P[y * N + z];
    }
    wave_fence();
    // lane i contracts the x-lines of row y = i (all z), as the middle phase does for its z-mode
#pragma unroll
    for (int z = 0; z < N; ++z) {
      double x[N], r[N];
#pragma unroll
      for (int j = 0; j < N; ++j) x[j] = cb[j * 33 + i * N + z];
      eo5(c, x, r);
#pragma unroll
      for (int j = 0; j < N; ++j) P[j * N + z] = r[j] * 0.2; // (lane i now holds the contracted row as [x][z])
    }
    wave_fence();
  }
  double s = 0;
  for (int e = 0; e < NN; ++e) s += P[e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__device__ __forceinline__ double bperm(int src_lane, double v)
{
  const int lo = __builtin_amdgcn_ds_bpermute(src_lane * 4, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(src_lane * 4, __double2hiint(v));
  return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(256, 2) void k_bpermute(double *out, int iters)
{
  const int lane = threadIdx.x & 63;
  const int g = lane / N, i = lane % N, base = g * N;
  double A[N];
  for (int j = 0; j < N; ++j) A[j] = 0.1 + 0.01 * (i * N + j); // this lane's row of the matrix
  double P[NN];
  for (int e = 0; e < NN; ++e) P[e] = 1.0 + 1e-3 * e + 1e-6 * threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int e = 0; e < NN; ++e) {
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const int src = base + j < 64 ? base + j : lane;
        acc = fma(A[j], bperm(src, P[e]), acc);
      }
      P[e] = acc * 0.2;
    }
  }
  double s = 0;
  for (int e = 0; e < NN; ++e) s += P[e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CTRL> __device__ __forceinline__ double dpp_shift(double v)
{
  // bound_ctrl: lanes shifted in from outside the row read 0
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(256, 2) void k_dpp(double *out, int iters)
{
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, i = r % N; // three groups per 16-lane row, lane 15 idle
  // coefficient of the value s lanes to the right / left: A[i][i+s], 0 outside the group
  double cs[9];
  for (int s = -4; s <= 4; ++s) cs[s + 4] = (i + s >= 0 && i + s < N && r < 15) ? 0.1 + 0.01 * (i * N + i + s) : 0.0;
  double P[NN];
  for (int e = 0; e < NN; ++e) P[e] = 1.0 + 1e-3 * e + 1e-6 * threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int e = 0; e < NN; ++e) {
      const double v = P[e];
      double acc = cs[4] * v;
      // row_shr:n = 0x110 + n (lane l reads lane l - n), row_shl:n = 0x100 + n (lane l reads lane l + n)
      acc = fma(cs[3], dpp_shift<0x111>(v), acc);
      acc = fma(cs[2], dpp_shift<0x112>(v), acc);
      acc = fma(cs[1], dpp_shift<0x113>(v), acc);
      acc = fma(cs[0], dpp_shift<0x114>(v), acc);
      acc = fma(cs[5], dpp_shift<0x101>(v), acc);
      acc = fma(cs[6], dpp_shift<0x102>(v), acc);
      acc = fma(cs[7], dpp_shift<0x103>(v), acc);
      acc = fma(cs[8], dpp_shift<0x104>(v), acc);
      P[e] = acc * 0.2;
    }
  }
  double s = 0;
  for (int e = 0; e < NN; ++e) s += P[e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <class K> static int run(const char *name, K kernel, int groups_per_wave, const char *counts)
{
  int dev = 0;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, dev));
  const int blocks = prop.multiProcessorCount * 2, iters = 2000;
  double *out;
  CK(hipMalloc(&out, size_t(blocks) * 256 * sizeof(double)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, 10);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  // every wave does `iters` plane contractions of `groups_per_wave` cell lines (25 values per lane each)
  const double ns_per_plane = best * 1e6 / iters; // per wave, two waves per SIMD running concurrently
  printf("%-9s %8.1f ns per plane contraction and wave  (%d groups of 5 lanes per wave: %6.2f ns per group)   %s\n", name,
         ns_per_plane, groups_per_wave, ns_per_plane / groups_per_wave, counts);
  CK(hipFree(out));
  return 0;
}

int main()
{
  printf("fp64 contraction of 5-lane groups, 25 values per lane, 256 threads x 2 workgroups per CU (two waves per SIMD)\n");
  if (run("lds", k_lds, 12, "25 ds_write_b64 + 25 ds_read_b64 + 85 fp64 VALU per lane")) return 1;
  if (run("bpermute", k_bpermute, 12, "250 ds_bpermute_b32 + 150 fp64 VALU per lane")) return 1;
  if (run("dpp", k_dpp, 12, "400 v_mov_b32_dpp + 250 fp64 VALU per lane (15 of 16 lanes)")) return 1;
  return 0;
}
