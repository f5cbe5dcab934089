// LDS lookup-rate microbenchmark: ds_read_b32 vs ds_read_b64 with lane-dependent (conflict-free) addresses
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int W>  // W = 1: b32, 2: b64
__global__ void __launch_bounds__(512, 8) k(float* out, int iters, unsigned seed) {
  __shared__ __attribute__((aligned(16))) float lds[8192];  // 32 KB
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 8192; i += 512) lds[i] = (float)i;
  __syncthreads();
  // entry rows of 64 lanes x W dwords: address = row * 64 * W * 4 + lane * W * 4 : conflict-free
  unsigned idx = seed ^ (tid * 2654435761u);
  float acc = 0.f, acc2 = 0.f;
  const unsigned boff = lane * 4 * W;  // the __shared__ array is the kernel's only LDS object: LDS address 0
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      // fixed rows (immediate offsets): no address arithmetic, the loop is LDS-bound
      constexpr unsigned rows[16] = {3, 11, 6, 14, 1, 9, 12, 4, 15, 7, 2, 10, 5, 13, 0, 8};
      const unsigned row = rows[u] ^ (idx & 0u);
      if (W == 1) acc += *reinterpret_cast<const volatile float __attribute__((address_space(3)))*>(boff + row * 256);
      else { f32x2 v = *reinterpret_cast<const volatile f32x2 __attribute__((address_space(3)))*>(boff + row * 512); acc += v.x; acc2 += v.y; }
    }
  }
  out[blockIdx.x * 512 + tid] = acc + acc2;
}
int main() {
  float* out; hipMalloc(&out, 4096 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int W = 1; W <= 2; ++W) for (int blocks : {256, 512, 1024}) {
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (W == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, out, iters, 7u);
      else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(512), 0, 0, out, iters, 7u);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) {
        double lane_reads = (double)blocks * 512 * iters * 16;
        printf("W=%d blocks=%d: %.3f ms, %.2f T lane-reads/s, %.2f T dwords/s (%.1f lanes/clk/CU at 2.4 GHz, 256 CUs)\n", W, blocks, ms,
               lane_reads / ms / 1e9, lane_reads * W / ms / 1e9, lane_reads / (ms * 1e-3) / 2.4e9 / 256);
      }
    }
  }
  return 0;
}
