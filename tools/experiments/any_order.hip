// any_order.hip -- does hipExtAnyOrderLaunch let the NEXT kernel of a stream start before the previous one has
// finished on gfx950 (hip_ext.h says "not supported on AMD GFX9xx boards")?  K1 spins ~200 us on a few workgroups,
// K2 stamps the 100 MHz clock at its start; both stamp their ends.
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/any_order.hip -o /tmp/any_order && /tmp/any_order
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>

__global__ void spin(unsigned long long* t, int us) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(10);
  if (threadIdx.x == 0 && blockIdx.x == 0) t[1] = __builtin_amdgcn_s_memrealtime();
}
__global__ void stamp(unsigned long long* t) {
  if (threadIdx.x == 0 && blockIdx.x == 0) t[2] = __builtin_amdgcn_s_memrealtime();
}

int main() {
  unsigned long long* t;
  hipMalloc(&t, 64);
  hipStream_t s;
  hipStreamCreate(&s);
  for (int flags = 0; flags < 2; ++flags) {
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(t, 0, 64);
      hipDeviceSynchronize();
      hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, nullptr, nullptr, 0, t, 200);
      hipExtLaunchKernelGGL(stamp, dim3(64), dim3(256), 0, s, nullptr, nullptr, flags, t);
      hipStreamSynchronize(s);
      unsigned long long h[3];
      hipMemcpy(h, t, 24, hipMemcpyDeviceToHost);
      printf("flags=%d: K1 start 0, K1 end %+.1f us, K2 start %+.1f us -> %s\n", flags, (h[1] - h[0]) / 100.0, (double)((long long)(h[2] - h[0])) / 100.0,
             h[2] < h[1] ? "K2 started BEFORE K1 ended (any-order honoured)" : "K2 waited for K1");
    }
  }
  return 0;
}
