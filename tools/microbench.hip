// Hardware questions behind the kernel design (MI355X / gfx950), each answered by a number:
//  1. v_fma_f64 issue rate; v_mfma_f64_16x16x4_f64 rate; do they overlap when issued by
//     different waves of one SIMD?
//  2. LDS ds_add_f64 rate vs ds_write_b64 (for in-workgroup accumulation of shared DoFs)
//  3. global_atomic_add_f64 rate (contiguous rows, as in the atomic scatter variant)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));

// mode 0: all waves VALU fma; mode 1: all waves MFMA; mode 2: even waves VALU, odd waves MFMA
__global__ __launch_bounds__(512) void fp64_rate(double *out, int iters, int mode)
{
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = mode == 1 || (mode == 2 && (wave & 1));
  double a = 1.0 + threadIdx.x * 1e-9, b = 0.999999;
  if (do_mfma) {
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  } else {
    double x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        x0 = fma(x0, b, a); x1 = fma(x1, b, a); x2 = fma(x2, b, a); x3 = fma(x3, b, a);
        x4 = fma(x4, b, a); x5 = fma(x5, b, a); x6 = fma(x6, b, a); x7 = fma(x7, b, a);
      }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  }
}

// mode 0: ds_write_b64; 1: ds_add_f64 (no return); 2: ds_read_b64
__global__ __launch_bounds__(256) void lds_rate(double *out, int iters, int mode)
{
  __shared__ double buf[256 * 9];
  for (int i = threadIdx.x; i < 256 * 9; i += 256) buf[i] = 0.0;
  __syncthreads();
  double v = threadIdx.x, acc = 0;
  double *p = buf + threadIdx.x;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (mode == 0) p[u * 256] = v + u;
      else if (mode == 1) atomicAdd(p + u * 256, v);
      else acc += p[u * 256];
    }
    if (mode == 2) __asm__ volatile("" ::: "memory");
  }
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = buf[threadIdx.x] + acc;
}

// every wave adds `rowlen` contiguous doubles per row at pseudo-random rows
__global__ __launch_bounds__(256) void gatomic_rate(double *dst, long nrows, int rowlen, int iters, int plain)
{
  const int lane = threadIdx.x & 63;
  long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  for (int i = 0; i < iters; ++i) {
    w = (w * 6364136223846793005L + 1442695040888963407L);
    long row = (unsigned long)(w >> 17) % (unsigned long)nrows;
    if (lane < rowlen) {
      if (plain) dst[row * rowlen + lane] = 1.0;
      else unsafeAtomicAdd(dst + row * rowlen + lane, 1.0);
    }
  }
}

int main()
{
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, clock %.0f MHz\n", prop.gcnArchName, cus, prop.clockRate / 1000.0);
  double *out;
  CK(hipMalloc(&out, sizeof(double) * 4096 * 512));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  for (int mode = 0; mode < 3; ++mode) {
    const int iters = 20000, blocks = cus * 2; // 8 waves per block -> 4 waves/SIMD
    hipLaunchKernelGGL(fp64_rate, dim3(blocks), dim3(512), 0, 0, out, 100, mode);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(fp64_rate, dim3(blocks), dim3(512), 0, 0, out, iters, mode);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    const double waves = blocks * 8.0;
    const double valu_waves = mode == 0 ? waves : (mode == 1 ? 0 : waves / 2), mfma_waves = waves - valu_waves;
    const double fl_valu = valu_waves * 64.0 * iters * 32 * 2, fl_mfma = mfma_waves * iters * 4.0 * 16 * 16 * 4 * 2;
    printf("fp64 mode %d (%s): %.3f ms  VALU %.1f TF/s  MFMA %.1f TF/s  total %.1f TF/s\n", mode,
           mode == 0 ? "VALU only" : mode == 1 ? "MFMA only" : "VALU+MFMA on alternating waves", ms,
           fl_valu / ms / 1e9, fl_mfma / ms / 1e9, (fl_valu + fl_mfma) / ms / 1e9);
  }
  for (int mode = 0; mode < 3; ++mode) {
    const int iters = 4000, blocks = cus * 8;
    hipLaunchKernelGGL(lds_rate, dim3(blocks), dim3(256), 0, 0, out, 10, mode);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(lds_rate, dim3(blocks), dim3(256), 0, 0, out, iters, mode);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = double(blocks) * 256 * iters * 8 * 8;
    printf("LDS mode %d (%s): %.3f ms  %.1f TB/s aggregate = %.1f B/clk/CU @2.4GHz\n", mode,
           mode == 0 ? "ds_write_b64" : mode == 1 ? "ds_add_f64" : "ds_read_b64", ms, bytes / ms / 1e9,
           bytes / ms / 1e-3 / cus / 2.4e9);
  }
  {
    const long nrows = 1 << 22; // x 25 doubles = 840 MB
    double *big;
    CK(hipMalloc(&big, sizeof(double) * nrows * 25));
    CK(hipMemset(big, 0, sizeof(double) * nrows * 25));
    for (int plain = 0; plain < 2; ++plain)
      for (int rowlen : {5, 25, 64}) {
        const int iters = 200, blocks = cus * 16;
        if (rowlen == 64) continue;
        hipLaunchKernelGGL(gatomic_rate, dim3(blocks), dim3(256), 0, 0, big, nrows, rowlen, 5, plain);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(gatomic_rate, dim3(blocks), dim3(256), 0, 0, big, nrows, rowlen, iters, plain);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = double(blocks) * 4 * iters * rowlen * 8;
        printf("global %s rows of %d doubles: %.3f ms  %.2f TB/s  (%.1f ns per wave-instruction per CU)\n",
               plain ? "plain store" : "atomic add f64", rowlen, ms, bytes / ms / 1e9,
               ms * 1e6 / (double(blocks) * 4 * iters / cus));
      }
    CK(hipFree(big));
  }
  return 0;
}
