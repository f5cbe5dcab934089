// lds_atomic_sparse.hip -- what does an LDS float atomic cost when only a few lanes of the wave are active?
// (round 2 measured ds_add_f32 with 64 active lanes on 64 addresses at 80 ns per wave instruction: lane by lane.
// The CSR role issues them with 2-3 active lanes -- the last lanes of row segments.)
// One workgroup of 512 threads per CU, every wave loops over `iters` blocks of 8 instructions with lanes
// [0, k) x stride active; reported: ns per wave instruction and CU (all 8 waves share the CU's LDS pipe).
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/lds_atomic_sparse.hip -o /tmp/lds_atomic_sparse && /tmp/lds_atomic_sparse
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int KIND>
__global__ void __launch_bounds__(512) k(float* out, int iters, int active, int spread) {
  __shared__ float lds[4096];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4096; i += 512) lds[i] = 0.f;
  __syncthreads();
  // active lanes: `active` of them, `spread` lanes apart
  const bool on = (lane % spread == 0) && (lane / spread < active);
  const uint32_t addr = 4u * ((tid * 7) & 4095);
  float v = 1.f + lane;
  if (on) {
    for (int it = 0; it < iters; ++it) {
      if (KIND == 0) asm volatile("ds_add_f32 %0, %1\n ds_add_f32 %0, %1 offset:4\n ds_add_f32 %0, %1 offset:8\n ds_add_f32 %0, %1 offset:12\n"
                                  "ds_add_f32 %0, %1 offset:16\n ds_add_f32 %0, %1 offset:20\n ds_add_f32 %0, %1 offset:24\n ds_add_f32 %0, %1 offset:28\n" ::"v"(addr), "v"(v) : "memory");
      if (KIND == 1) asm volatile("ds_add_u32 %0, %1\n ds_add_u32 %0, %1 offset:4\n ds_add_u32 %0, %1 offset:8\n ds_add_u32 %0, %1 offset:12\n"
                                  "ds_add_u32 %0, %1 offset:16\n ds_add_u32 %0, %1 offset:20\n ds_add_u32 %0, %1 offset:24\n ds_add_u32 %0, %1 offset:28\n" ::"v"(addr), "v"(v) : "memory");
      if (KIND == 2) asm volatile("ds_write_b32 %0, %1\n ds_write_b32 %0, %1 offset:4\n ds_write_b32 %0, %1 offset:8\n ds_write_b32 %0, %1 offset:12\n"
                                  "ds_write_b32 %0, %1 offset:16\n ds_write_b32 %0, %1 offset:20\n ds_write_b32 %0, %1 offset:24\n ds_write_b32 %0, %1 offset:28\n" ::"v"(addr), "v"(v) : "memory");
      if (KIND == 3) asm volatile("ds_add_rtn_f32 %1, %0, %1\n s_waitcnt lgkmcnt(0)\n ds_add_rtn_f32 %1, %0, %1 offset:4\n s_waitcnt lgkmcnt(0)\n"
                                  "ds_add_rtn_f32 %1, %0, %1 offset:8\n s_waitcnt lgkmcnt(0)\n ds_add_rtn_f32 %1, %0, %1 offset:12\n s_waitcnt lgkmcnt(0)\n"
                                  "ds_add_rtn_f32 %1, %0, %1 offset:16\n s_waitcnt lgkmcnt(0)\n ds_add_rtn_f32 %1, %0, %1 offset:20\n s_waitcnt lgkmcnt(0)\n"
                                  "ds_add_rtn_f32 %1, %0, %1 offset:24\n s_waitcnt lgkmcnt(0)\n ds_add_rtn_f32 %1, %0, %1 offset:28\n s_waitcnt lgkmcnt(0)\n" : : "v"(addr), "v"(v) : "memory");
      if (KIND == 4) asm volatile("ds_read_b32 %1, %0\n ds_read_b32 %1, %0 offset:4\n ds_read_b32 %1, %0 offset:8\n ds_read_b32 %1, %0 offset:12\n"
                                  "ds_read_b32 %1, %0 offset:16\n ds_read_b32 %1, %0 offset:20\n ds_read_b32 %1, %0 offset:24\n ds_read_b32 %1, %0 offset:28\n s_waitcnt lgkmcnt(0)\n" : : "v"(addr), "v"(v) : "memory");
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) out[blockIdx.x] = lds[0] + v;
}

template <int KIND>
static void run(const char* name, float* out) {
  const int iters = 2000;
  const int acts[][2] = {{64, 1}, {32, 2}, {16, 4}, {8, 8}, {4, 16}, {2, 32}, {1, 64}, {4, 1}, {2, 1}};
  for (auto& a : acts) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<KIND><<<256, 512>>>(out, 10, a[0], a[1]);
    CHECK(hipEventRecord(e0));
    k<KIND><<<256, 512>>>(out, iters, a[0], a[1]);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    // per CU: 8 waves x iters x 8 instructions
    printf("%-16s active %2d lanes (every %2d): %7.2f ns per wave instruction and CU\n", name, a[0], a[1], ms * 1e6 / (8.0 * iters * 8));
  }
}

int main() {
  float* out; CHECK(hipMalloc(&out, 4096));
  run<0>("ds_add_f32", out); run<1>("ds_add_u32", out); run<2>("ds_write_b32", out); run<3>("ds_add_rtn_f32+wait", out); run<4>("ds_read_b32", out);
  return 0;
}
