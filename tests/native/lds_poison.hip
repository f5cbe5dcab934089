// lds_poison.hip -- test infrastructure (tests/conftest.py), not product: fill the LDS of EVERY compute unit with a poison
// pattern, so that a kernel which reads an LDS word before writing it (a ticket it assumes zero, a slab it assumes clean)
// picks up NaN / 1e38 instead of whatever the previous kernel on that CU happened to leave there.  LDS is not cleared
// between workgroups or kernels; a workgroup that declares all 160 KB has a CU's whole LDS to itself, and eight of them
// per CU in the grid reach every CU whatever the dispatch order.
#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr int kLdsBytes = 160 * 1024;

__global__ void __launch_bounds__(1024) lds_poison_kernel(uint32_t pattern, uint32_t* sink) {
  extern __shared__ uint32_t lds[];
  for (int i = threadIdx.x; i < kLdsBytes / 4; i += 1024) lds[i] = pattern;
  __syncthreads();
  // (read something back so that the stores cannot be dropped)
  if (lds[(threadIdx.x * 97u) % (kLdsBytes / 4)] != pattern) sink[0] = 1u;
}

extern "C" int lds_poison(void* stream, uint32_t pattern, uint32_t* sink) {
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lds_poison_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    if (e != hipSuccess) return (int)e;
    configured = true;
  }
  int dev = 0, cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
  hipLaunchKernelGGL(lds_poison_kernel, dim3(8 * cus), dim3(1024), kLdsBytes, static_cast<hipStream_t>(stream), pattern, sink);
  return (int)hipGetLastError();
}
