// icache_cold.hip -- what does a COLD instruction cache cost at the start of a launch?
// (Round 3: every kernel of this repo carries ~3 us per launch over a loads-only calibration kernel of a few hundred
// bytes; the fused kernels execute ~10 KB of straight-line code per wave.  Instruction caches are invalidated at
// every dispatch.)  Each wave runs the SAME block of N 8-byte VALU instructions twice -- a loop, same addresses --
// and stamps the constant 100 MHz clock around each pass: pass 1 fetches the block from L2, pass 2 hits.
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/icache_cold.hip -o build/exp/icache_cold && build/exp/icache_cold
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef unsigned long long u64;

#define STR2(x) #x
#define STR(x) STR2(x)

template <int N>
__global__ void __launch_bounds__(512) k_code(u64* out, int reps) {
  unsigned a = threadIdx.x, b = blockIdx.x;
  u64 t[3];
  t[0] = __builtin_amdgcn_s_memrealtime();
  for (int r = 0; r < reps; ++r) {
    if constexpr (N == 256) asm volatile(".rept 256\n v_mad_u32_u24 %0, %0, %1, %1\n .endr" : "+v"(a) : "v"(b));
    if constexpr (N == 1024) asm volatile(".rept 1024\n v_mad_u32_u24 %0, %0, %1, %1\n .endr" : "+v"(a) : "v"(b));
    if constexpr (N == 2048) asm volatile(".rept 2048\n v_mad_u32_u24 %0, %0, %1, %1\n .endr" : "+v"(a) : "v"(b));
    if constexpr (N == 4096) asm volatile(".rept 4096\n v_mad_u32_u24 %0, %0, %1, %1\n .endr" : "+v"(a) : "v"(b));
    asm volatile("s_nop 0" ::: "memory");
    if (r < 2) t[r + 1] = __builtin_amdgcn_s_memrealtime();
  }
  if ((threadIdx.x & 63) == 0) {
    u64* o = out + 4 * ((size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64);
    o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = a;
  }
}

template <int N>
static void run(int grid, int threads, u64* d_out) {
  const int waves = grid * threads / 64;
  std::vector<u64> h(4 * (size_t)waves);
  double p1 = 0, p2 = 0, p1max = 0, span = 0;
  const int reps = 5;
  for (int r = 0; r < reps + 1; ++r) {
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_code<N>, dim3(grid), dim3(threads), 0, 0, d_out, 2);
    CHECK(hipDeviceSynchronize());
    if (r == 0) continue;
    CHECK(hipMemcpy(h.data(), d_out, h.size() * sizeof(u64), hipMemcpyDeviceToHost));
    u64 tmin = ~0ull, tmax = 0;
    double s1 = 0, s2 = 0, m1 = 0;
    for (int i = 0; i < waves; ++i) {
      const double a = (double)(h[4 * i + 1] - h[4 * i]) / 100.0, b = (double)(h[4 * i + 2] - h[4 * i + 1]) / 100.0;
      s1 += a; s2 += b; m1 = std::max(m1, a);
      tmin = std::min(tmin, h[4 * i]); tmax = std::max(tmax, h[4 * i + 2]);
    }
    p1 += s1 / waves; p2 += s2 / waves; p1max += m1; span += (double)(tmax - tmin) / 100.0;
  }
  printf("%5d instructions (%5.1f KB) grid=%4d x %4d threads: pass 1 (cold) mean %6.2f us, max %6.2f us; pass 2 (warm) mean %6.2f us; first entry -> last exit %6.2f us\n",
         N, N * 8 / 1024.0, grid, threads, p1 / reps, p1max / reps, p2 / reps, span / reps);
  fflush(stdout);
}

int main() {
  u64* d_out;
  CHECK(hipMalloc(&d_out, 4 * 16384 * sizeof(u64)));
  for (int threads : {64, 512}) {
    for (int grid : {256, 1024}) {
      run<256>(grid, threads, d_out);
      run<1024>(grid, threads, d_out);
      run<2048>(grid, threads, d_out);
      run<4096>(grid, threads, d_out);
    }
  }
  return 0;
}
