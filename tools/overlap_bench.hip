// Does LDS traffic of one wave overlap with fp64 VALU work of another wave on the same SIMD?
// Each wave alternates a VALU segment (NV independent-chain fp64 FMAs) with an LDS segment
// (NL ds_write_b64 + NL ds_read_b64 on a wave-private slab); odd waves start with the LDS
// segment.  Reports time for VALU only, LDS only, both.  (MI355X design question of DESIGN.md 6.)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e)); return 1; } } while (0)

template <int NV, int NL>
__global__ __launch_bounds__(256) void k(double *out, int iters, int mode)
{
  __shared__ double lds[4 * 64 * 26];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double *slab = lds + wave * 64 * 26;
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = 1.0 + i + lane * 1e-9;
  const double b = 0.999999, c = 1e-7;
  double r[NL > 0 ? NL : 1];
  for (int i = 0; i < NL; ++i) r[i] = lane + i;
  const bool lds_first = (wave + blockIdx.x) & 1;
  for (int it = 0; it < iters; ++it) {
    for (int half = 0; half < 2; ++half) {
      const bool do_lds = (half == 0) == lds_first;
      if (do_lds) {
        if (mode & 2) {
#pragma unroll
          for (int i = 0; i < NL; ++i) slab[i * 64 + lane] = r[i];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
          for (int i = 0; i < NL; ++i) r[i] = slab[i * 64 + ((lane + 5 * i + 1) & 63)];
        }
      } else if (mode & 1) {
#pragma unroll
        for (int i = 0; i < NV / 8; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) a[j] = fma(a[j], b, c);
      }
    }
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  for (int i = 0; i < NL; ++i) s += r[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, int NL> int run(double *out, int wg_per_cu)
{
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2000, blocks = 256 * wg_per_cu;
  float ms[4];
  for (int mode = 1; mode <= 3; ++mode) {
    hipLaunchKernelGGL((k<NV, NL>), dim3(blocks), dim3(256), 0, 0, out, 10, mode);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<NV, NL>), dim3(blocks), dim3(256), 0, 0, out, iters, mode);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[mode], e0, e1));
  }
  printf("NV=%4d NL=%3d  %d WG/CU (%d waves/SIMD):  VALU %.3f ms  LDS %.3f ms  both %.3f ms  (sum %.3f, max %.3f)\n", NV, NL, wg_per_cu,
         wg_per_cu, ms[1], ms[2], ms[3], ms[1] + ms[2], ms[1] > ms[2] ? ms[1] : ms[2]);
  return 0;
}

int main()
{
  double *out;
  CK(hipMalloc(&out, sizeof(double) * 256 * 256 * 8));
  for (int w = 1; w <= 4; ++w) run<400, 25>(out, w);
  for (int w = 1; w <= 4; ++w) run<400, 50>(out, w);
  for (int w = 2; w <= 3; ++w) run<1200, 100>(out, w);
  return 0;
}
