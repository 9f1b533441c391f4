// What does one vector-memory wave-instruction cost a CU, as a function of its access shape?
// (MI355X / gfx950.)  Every workgroup streams through its own region; 8 waves per CU as in the tile
// kernel (2 workgroups x 4 waves).  Output: ns per wave-instruction per CU and TB/s.
//   shape 0: dwordx4, 64 lanes contiguous, 1 KB aligned
//   shape 1: dwordx2, 64 lanes contiguous, 512 B aligned
//   shape 2: dwordx2, two rows of 25 doubles (lanes 0..24 / 32..56), row stride 289 doubles  [tile store phase]
//   shape 3: dwordx4 gather of the tile kernel: lane = (cell 0..5, block 0..1, plane 0..4):
//            16 B at block*B + plane*289*289 + 4*cell doubles, rows 8-B aligned              [tile src gather]
//   shape 4: dwordx4, four rows of 16 lanes (256 B each), row stride 289 doubles, 8-B aligned
//   shape 5: dwordx2, one row of 49 doubles (lanes 0..48), row stride 289 doubles
//   shape 6: dwordx4, lanes 0..24 cover one row of 50 doubles = two adjacent 25-double rows? no: one 400-B row
//   shape 7: dwordx2, 64 lanes contiguous but 8-B (not 16-B) aligned start
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr long NX = 289, PLANE = NX * NX;

template <int SHAPE, bool STORE>
__global__ __launch_bounds__(256) void vmem(double *buf, long region, int iters, double *sink)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double *base = buf + blockIdx.x * region;
  double acc = 0;
  d2 acc2 = {0, 0};
  for (int it = 0; it < iters; ++it) {
    long off;
    bool active = true;
    int width = 2; // doubles per lane
    if (SHAPE == 0) { off = (long(it) * 4 + wave) * 128 + lane * 2; }
    else if (SHAPE == 1) { off = (long(it) * 4 + wave) * 64 + lane; width = 1; }
    else if (SHAPE == 7) { off = (long(it) * 4 + wave) * 64 + lane + 1; width = 1; }
    else if (SHAPE == 2) { // rows 2*wave + (lane>>5) + 8*it of a plane-like region
      const long row = 8L * it + 2 * wave + (lane >> 5);
      off = row * NX + (lane & 31); active = (lane & 31) < 25; width = 1;
    } else if (SHAPE == 5) {
      const long row = 4L * it + wave;
      off = row * NX + lane; active = lane < 49; width = 1;
    } else if (SHAPE == 3) {
      const int k = lane % 5, b = (lane / 5) % 2, c = lane / 10;
      active = lane < 60;
      // one (row y, x-part) of the gather per iteration; rows advance with it, waves are cell rows
      const long row = 4L * wave + (it % 5) + 16L * (it / 15);
      off = b * (10 * PLANE) + k * PLANE + row * NX + 4 * (c % 6) + 2 * ((it / 5) % 3 == 2 ? 2 : (it / 5) % 3);
    } else if (SHAPE == 4) {
      const long row = 16L * it + 4 * wave + (lane >> 4);
      off = row * NX + 2 * (lane & 15) + 1;
    } else { // 6
      const long row = 4L * it + wave;
      off = row * NX + 2 * lane + 1; active = lane < 25;
    }
    if (!active) continue;
    double *p = base + off;
    if (STORE) {
      if (width == 2) { d2 v = {1.0, 2.0}; __builtin_nontemporal_store(v, (d2 *)p); }
      else __builtin_nontemporal_store(1.0, p);
    } else {
      if (width == 2) { d2 v; asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v) : "v"(p) : "memory"); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); acc2 += v; }
      else { double v; asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(v) : "v"(p) : "memory"); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); acc += v; }
    }
  }
  if (!STORE) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc + acc2.x + acc2.y == 1.2345e30) sink[0] = acc;
  }
}

template <int SHAPE, bool STORE> int run(double *buf, long region, double *sink, int cus, const char *name, double bytes_per_instr)
{
  const int blocks = cus * 2, iters = 600;
  printf("launch shape %d store %d\n", SHAPE, int(STORE));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  hipLaunchKernelGGL((vmem<SHAPE, STORE>), dim3(blocks), dim3(256), 0, 0, buf, region, 50, sink);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((vmem<SHAPE, STORE>), dim3(blocks), dim3(256), 0, 0, buf, region, iters, sink);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipDeviceSynchronize()); CK(hipGetLastError());
  const double instr = double(blocks) * 4 * iters;
  printf("%-5s shape %d %-58s %7.3f ms  %6.1f ns/instr/CU  %5.2f TB/s\n", STORE ? "store" : "load", SHAPE, name, ms,
         ms * 1e6 / (instr / cus), instr * bytes_per_instr / ms / 1e9);
  return 0;
}

int main()
{
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const long region = 12L * PLANE + 600L * 16 * NX + 4096; // doubles per workgroup (largest footprint: shape 3/4)
  double *buf, *sink;
  CK(hipMalloc(&buf, sizeof(double) * region * cus * 2));
  for (long o = 0; o < region * cus * 2; o += (1L << 27))
    CK(hipMemset(buf + o, 0, sizeof(double) * (region * cus * 2 - o < (1L << 27) ? region * cus * 2 - o : (1L << 27))));
  CK(hipDeviceSynchronize());
  CK(hipMalloc(&sink, 64));
  printf("buf %p sink %p region %ld\n", (void *)buf, (void *)sink, region);
  printf("device %s, %d CUs; %.1f GB buffer\n", prop.gcnArchName, cus, region * cus * 2 * 8 / 1e9);
#define BOTH(S, NAME, BYTES) run<S, false>(buf, region, sink, cus, NAME, BYTES); run<S, true>(buf, region, sink, cus, NAME, BYTES);
  BOTH(0, "dwordx4 contiguous 1 KB aligned", 1024.0)
  BOTH(1, "dwordx2 contiguous 512 B aligned", 512.0)
  BOTH(7, "dwordx2 contiguous 512 B, 8-B aligned only", 512.0)
  BOTH(2, "dwordx2 two rows of 25 doubles (tile store shape)", 400.0)
  BOTH(5, "dwordx2 one row of 49 doubles", 392.0)
  BOTH(4, "dwordx4 four rows of 256 B, 8-B aligned", 1024.0)
  BOTH(6, "dwordx4 one row of 400 B, 8-B aligned", 400.0)
  CK(hipDeviceSynchronize());
  return 0;
}
