// dispatch_ramp.hip -- how long does the chip take to START the workgroups of one launch, and what
// does that depend on?  (Round-2 review, "what's weak" item 2: 768 workgroups of the product kernel
// take 2.7 us to start; the loads-only calibration runs 1.3-2 us slower at 1024 workgroups than at
// 256.  Is the cost per workgroup, per wave, per resident round, or a consequence of the memory
// traffic the first workgroups generate?)
//
// Every workgroup stamps the 100 MHz constant clock (s_memrealtime) at its first instruction and at
// its end, and records its XCC / SE / CU ids.  Bodies:
//   mode 0  nothing
//   mode 1  stream: every lane issues NLOAD nontemporal 16-byte loads from a large buffer (each
//           workgroup its own contiguous piece), waits, folds
//   mode 2  spin for SPIN ticks of the constant clock (a resident workgroup that holds its slot)
// Variables: workgroup size (256 / 512 / 1024 threads), grid size, dynamic LDS per workgroup (caps
// the workgroups resident per CU), VGPRs (template: a live register array).
//
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/dispatch_ramp.hip -o build/dispatch_ramp && build/dispatch_ramp
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

// s_getreg_b32 simm16 = id | offset << 6 | (size - 1) << 11
#define GETREG(id, off, size) __builtin_amdgcn_s_getreg((id) | ((off) << 6) | (((size) - 1) << 11))

template <int T, int NLOAD, int VPAD>
__global__ void __launch_bounds__(T) k_ramp(u64* out, const u32x4* __restrict__ buf, size_t n16, int mode, int spin, float* sink) {
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  uint32_t acc = 0;
  if (mode == 1) {
    // this workgroup's contiguous piece: NLOAD * T 16-byte words
    size_t base = ((size_t)blockIdx.x * NLOAD * T) % (n16 - (size_t)NLOAD * T);
    u32x4 w[NLOAD];
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) w[i] = __builtin_nontemporal_load(buf + base + (size_t)i * T + tid);
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) acc ^= w[i].x ^ w[i].y ^ w[i].z ^ w[i].w;
  } else if (mode == 2) {
    while ((int)(__builtin_amdgcn_s_memrealtime() - t0) < spin) __builtin_amdgcn_s_sleep(1);
  }
  if constexpr (VPAD > 0) {  // keep VPAD registers live across the body (raises the kernel's VGPR count)
    float v[VPAD];
#pragma unroll
    for (int i = 0; i < VPAD; ++i) v[i] = lds[(tid + i * 7) & 63] + (float)i;
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < VPAD; ++i) acc ^= __builtin_bit_cast(uint32_t, v[i]);
  }
  if (acc == 0x12345678u) sink[0] = 1.f;
  __syncthreads();
  if (tid == 0) {
    const u64 t1 = __builtin_amdgcn_s_memrealtime();
    const uint32_t hw = GETREG(4, 0, 32);   // HW_REG_HW_ID
    const uint32_t xcc = GETREG(20, 0, 4);  // HW_REG_XCC_ID
    out[4 * (size_t)blockIdx.x + 0] = t0;
    out[4 * (size_t)blockIdx.x + 1] = t1;
    out[4 * (size_t)blockIdx.x + 2] = hw;
    out[4 * (size_t)blockIdx.x + 3] = xcc;
  }
}

// ------------------------------------------------------------------------------------------------
// Dependent kernel-argument reads: the product kernel finds its segment, role and geometry through a
// chain of scalar loads from a ~600-byte by-value argument block, each waiting for the one before
// (block0[] -> s -> seg[s].gm.sparse_last -> dense_block0 -> ...).  What does one link cost for the first
// wave of a CU (scalar cache cold) and for the later ones?
// ------------------------------------------------------------------------------------------------
struct Chain { int next[160]; };  // 640 bytes by value; next[i] = the index to read after i

template <int LINKS>
__global__ void __launch_bounds__(512) k_chain(u64* out, const Chain c, int start) {
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  int i = start;
#pragma unroll
  for (int l = 0; l < LINKS; ++l) i = __builtin_amdgcn_readfirstlane(c.next[i]);  // dependent s_load per link
  const u64 t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    out[4 * (size_t)blockIdx.x + 0] = t0;
    out[4 * (size_t)blockIdx.x + 1] = t1 + (i == 12345 ? 1 : 0);
    out[4 * (size_t)blockIdx.x + 2] = GETREG(4, 0, 32);
    out[4 * (size_t)blockIdx.x + 3] = GETREG(20, 0, 4);
  }
}

template <int LINKS>
static void run_chain(int grid, u64* d_out) {
  Chain c;
  for (int i = 0; i < 160; ++i) c.next[i] = (i * 37 + 11) % 160;  // hops across the whole block (different cache lines)
  std::vector<u64> h(4 * (size_t)grid);
  double first_sum = 0, later_sum = 0, end_last = 0;
  int nfirst = 0, nlater = 0;
  const int reps = 5;
  for (int r = 0; r < reps + 1; ++r) {
    CHECK(hipMemset(d_out, 0, 4 * (size_t)grid * sizeof(u64)));
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_chain<LINKS>, dim3(grid), dim3(512), 0, 0, d_out, c, r % 160);
    CHECK(hipDeviceSynchronize());
    if (r == 0) continue;
    CHECK(hipMemcpy(h.data(), d_out, h.size() * sizeof(u64), hipMemcpyDeviceToHost));
    u64 tmin = ~0ull;
    for (int i = 0; i < grid; ++i) tmin = std::min(tmin, h[4 * i]);
    // the first workgroup (by entry time) on each CU vs the rest
    std::vector<u64> first_entry(8 * 64 * 4, ~0ull);
    auto cu_of = [&](int i) { const uint32_t hw = (uint32_t)h[4 * i + 2]; return ((((int)h[4 * i + 3] & 7) * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 15); };
    for (int i = 0; i < grid; ++i) first_entry[cu_of(i)] = std::min(first_entry[cu_of(i)], h[4 * i]);
    for (int i = 0; i < grid; ++i) {
      const double d = (double)(h[4 * i + 1] - h[4 * i]) / 100.0;
      if (h[4 * i] == first_entry[cu_of(i)]) { first_sum += d; ++nfirst; } else { later_sum += d; ++nlater; }
      end_last = std::max(end_last, (double)(h[4 * i + 1] - tmin) / 100.0);
    }
  }
  printf("kernarg chain of %2d dependent scalar loads, grid=%4d: first workgroup of a CU %5.2f us, later workgroups %5.2f us, last end %5.2f us\n",
         LINKS, grid, first_sum / std::max(nfirst, 1), nlater ? later_sum / nlater : 0.0, end_last / reps);
  fflush(stdout);
}

struct Result { double entry50, entry90, entry_last, end_last, ev_us; int cus_used; int max_per_cu; };

template <int T, int NLOAD, int VPAD>
static Result run(int grid, int lds_bytes, int mode, int spin, u64* d_out, const u32x4* buf, size_t n16, float* sink, int reps) {
  auto kern = k_ramp<T, NLOAD, VPAD>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256));
  std::vector<u64> h(4 * (size_t)grid);
  Result acc{};
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int r = 0; r < reps + 1; ++r) {
    CHECK(hipMemset(d_out, 0, 4 * (size_t)grid * sizeof(u64)));
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(T), lds_bytes, 0, d_out, buf, n16, mode, spin, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    if (r == 0) continue;  // warm
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipMemcpy(h.data(), d_out, h.size() * sizeof(u64), hipMemcpyDeviceToHost));
    u64 tmin = ~0ull;
    for (int i = 0; i < grid; ++i) tmin = std::min(tmin, h[4 * i]);
    std::vector<double> ent(grid);
    double end_last = 0;
    std::vector<int> percu(8 * 64 * 4, 0);
    for (int i = 0; i < grid; ++i) {
      ent[i] = (double)(h[4 * i] - tmin) / 100.0;
      end_last = std::max(end_last, (double)(h[4 * i + 1] - tmin) / 100.0);
      const uint32_t hw = (uint32_t)h[4 * i + 2];
      const int cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, xcc = (int)h[4 * i + 3] & 7;
      percu[((xcc * 8 + se) * 2 + sh) * 16 + cu]++;
    }
    std::sort(ent.begin(), ent.end());
    acc.entry50 += ent[grid / 2];
    acc.entry90 += ent[(size_t)(grid * 0.9)];
    acc.entry_last += ent[grid - 1];
    acc.end_last += end_last;
    acc.ev_us += ms * 1000.0;
    int used = 0, mx = 0;
    for (int v : percu) { used += v > 0; mx = std::max(mx, v); }
    acc.cus_used = used;
    acc.max_per_cu = mx;
  }
  acc.entry50 /= reps; acc.entry90 /= reps; acc.entry_last /= reps; acc.end_last /= reps; acc.ev_us /= reps;
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
  return acc;
}

static void report(const char* what, int T, int grid, int lds, const Result& r) {
  printf("%-34s T=%4d grid=%5d lds=%6d : entry p50 %5.2f p90 %5.2f last %5.2f us | last end %6.2f us | events %6.2f us | CUs %3d, max WGs seen on one CU %d\n",
         what, T, grid, lds, r.entry50, r.entry90, r.entry_last, r.end_last, r.ev_us, r.cus_used, r.max_per_cu);
  fflush(stdout);
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs\n", prop.name, prop.multiProcessorCount);
  const size_t n16 = (size_t)256 << 20 >> 4;  // 256 MiB of 16-byte words
  u32x4* buf;
  CHECK(hipMalloc(&buf, n16 * 16));
  CHECK(hipMemset(buf, 0x5a, n16 * 16));
  u64* d_out;
  CHECK(hipMalloc(&d_out, 4 * 8192 * sizeof(u64)));
  float* sink;
  CHECK(hipMalloc(&sink, 4));
  const int reps = 5;
  if (getenv("RAMP_CHAIN_ONLY") || true) {
    for (int g : {256, 1024}) { run_chain<1>(g, d_out); run_chain<2>(g, d_out); run_chain<4>(g, d_out); run_chain<8>(g, d_out); }
    if (getenv("RAMP_CHAIN_ONLY")) return 0;
  }
  const int grids[] = {256, 512, 768, 1024, 1536, 2048};
  // 1. empty bodies: pure dispatch, by workgroup size
  for (int g : grids) report("empty", 256, g, 0, run<256, 1, 0>(g, 0, 0, 0, d_out, buf, n16, sink, reps));
  for (int g : grids) report("empty", 512, g, 0, run<512, 1, 0>(g, 0, 0, 0, d_out, buf, n16, sink, reps));
  for (int g : grids) if (g <= 1024) report("empty", 1024, g, 0, run<1024, 1, 0>(g, 0, 0, 0, d_out, buf, n16, sink, reps));
  // 2. resident workgroups (spin 4 us): does the second / third / fourth workgroup of a CU start later?
  for (int g : {256, 512, 768, 1024}) report("spin 4us", 512, g, 0, run<512, 1, 0>(g, 0, 2, 400, d_out, buf, n16, sink, reps));
  for (int g : {256, 512}) report("spin 4us", 1024, g, 0, run<1024, 1, 0>(g, 0, 2, 400, d_out, buf, n16, sink, reps));
  for (int g : {512, 1024, 2048}) report("spin 4us", 256, g, 0, run<256, 1, 0>(g, 0, 2, 400, d_out, buf, n16, sink, reps));
  // 3. with LDS (10 KB and 36 KB per workgroup, the product kernels' footprints) and 56 live VGPRs
  for (int g : {256, 512, 768, 1024}) report("spin 4us, 10 KB lds", 512, g, 10240, run<512, 1, 0>(g, 10240, 2, 400, d_out, buf, n16, sink, reps));
  for (int g : {256, 512, 768, 1024}) report("spin 4us, 36 KB lds", 512, g, 36864, run<512, 1, 0>(g, 36864, 2, 400, d_out, buf, n16, sink, reps));
  for (int g : {256, 512, 768, 1024}) report("spin 4us, 10 KB lds, 48 vgpr pad", 512, g, 10240, run<512, 1, 48>(g, 10240, 2, 400, d_out, buf, n16, sink, reps));
  // 4. streaming bodies: 4 / 8 loads of 16 bytes per lane
  for (int g : grids) report("stream 4 x 16 B per lane", 512, g, 10240, run<512, 4, 0>(g, 10240, 1, 0, d_out, buf, n16, sink, reps));
  for (int g : {256, 512, 1024}) report("stream 8 x 16 B per lane", 512, g, 10240, run<512, 8, 0>(g, 10240, 1, 0, d_out, buf, n16, sink, reps));
  for (int g : {256, 512}) report("stream 4 x 16 B per lane", 1024, g, 10240, run<1024, 4, 0>(g, 10240, 1, 0, d_out, buf, n16, sink, reps));
  for (int g : {256, 512}) report("stream 8 x 16 B per lane", 1024, g, 10240, run<1024, 8, 0>(g, 10240, 1, 0, d_out, buf, n16, sink, reps));
  for (int g : {512, 1024, 2048, 4096}) report("stream 4 x 16 B per lane", 256, g, 10240, run<256, 4, 0>(g, 10240, 1, 0, d_out, buf, n16, sink, reps));
  return 0;
}
