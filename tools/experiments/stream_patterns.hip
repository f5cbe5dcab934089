// stream_patterns.hip -- how fast can ONE workgroup per CU pull a row-major int32 [rows, N] matrix
// out of HBM, by access pattern?  (Calibration for the v2 dense kernel: its workgroup reads a
// 256-byte-wide column strip, one dword per lane.)  Nothing is computed: words are xor-ed.
//
//   pattern 0  strip, 64 columns wide (256 B), one dword per lane; waves interleave rows
//   pattern 1  strip, 256 columns wide (1 KiB), dwordx4 per lane; waves interleave rows
//   pattern 2  linear: each workgroup reads one contiguous range, dwordx4 per lane
//   pattern 3  strip 64 wide, but a wave owns a CONTIGUOUS block of rows (not interleaved)
//   pattern 4  pattern 0 through buffer_load_dword (a descriptor over the matrix) instead of global_load_dword
//   pattern 5  pattern 4 with a ROLLING ring: two loads issued per two words consumed (D words in flight)
// Workgroups take contiguous ranges of the flattened (strip, row) space (as the v2 kernel does).
// Every variant: T waves per workgroup, D loads in flight per wave (issue D, consume D, repeat, two
// register sets so that D..2D are outstanding), grid = G workgroups per CU.
//
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/stream_patterns.hip -o /tmp/sp && /tmp/sp
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int VEC> struct Word;
template <> struct Word<1> { using type = uint32_t; };
template <> struct Word<4> { using type = u32x4; };
__device__ __forceinline__ uint32_t fold(uint32_t v) { return v; }
__device__ __forceinline__ uint32_t fold(u32x4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

template <int PATTERN, int T, int D, bool NT>
__global__ void __launch_bounds__(T * 64) k_stream(const uint32_t* __restrict__ q, int rows, int N, int units_per_wg, float* y) {
  constexpr int VEC = (PATTERN == 0 || PATTERN == 3) ? 1 : 4;
  using W = typename Word<VEC>::type;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint32_t acc = 0;
  if constexpr (PATTERN == 2) {
    // contiguous: units = 1 KiB blocks (64 lanes x 16 B)
    const size_t total = (size_t)rows * N / 256;
    size_t g = (size_t)blockIdx.x * units_per_wg, g_end = g + units_per_wg;
    if (g_end > total) g_end = total;
    const u32x4* p = reinterpret_cast<const u32x4*>(q);
    for (size_t u = g + w; u < g_end; u += 2 * T * D) {
      u32x4 a[D], b[D];
#pragma unroll
      for (int j = 0; j < D; ++j) { size_t uu = u + (size_t)j * T; if (uu > g_end - 1) uu = g_end - 1; a[j] = NT ? __builtin_nontemporal_load(p + uu * 64 + lane) : p[uu * 64 + lane]; }
#pragma unroll
      for (int j = 0; j < D; ++j) { size_t uu = u + (size_t)(D + j) * T; if (uu > g_end - 1) uu = g_end - 1; b[j] = NT ? __builtin_nontemporal_load(p + uu * 64 + lane) : p[uu * 64 + lane]; }
#pragma unroll
      for (int j = 0; j < D; ++j) acc ^= fold(a[j]);
#pragma unroll
      for (int j = 0; j < D; ++j) acc ^= fold(b[j]);
    }
  } else {
    const int strip_cols = 64 * VEC;
    const int n_strips = (N + strip_cols - 1) / strip_cols;
    const unsigned total = (unsigned)n_strips * rows;
    unsigned g = blockIdx.x * (unsigned)units_per_wg, g_end = g + units_per_wg;
    if (g_end > total) g_end = total;
    while (g < g_end) {
      const unsigned strip = g / rows;
      const int r0 = g - strip * rows;
      int r1 = rows;
      if ((unsigned)(r1 - r0) > g_end - g) r1 = r0 + (g_end - g);
      g += r1 - r0;
      int c = strip * strip_cols + lane * VEC;
      if (c > N - VEC) c = N - VEC;
      const char* base = reinterpret_cast<const char*>(q) + 4u * c;
      const uint32_t row_bytes = 4u * N;
      const int n = r1 - r0;
      int first, step, cnt;
      if (PATTERN == 3) { const int per = (n + T - 1) / T; first = r0 + w * per; step = 1; cnt = n - w * per; if (cnt > per) cnt = per; if (cnt < 0) cnt = 0; }
      else { first = r0 + w; step = T; cnt = n > w ? (n - w + T - 1) / T : 0; }
      for (int i = 0; i < cnt; i += 2 * D) {
        W a[D], b[D];
#pragma unroll
        for (int j = 0; j < D; ++j) { int ii = i + j; if (ii > cnt - 1) ii = cnt - 1; const W* p = reinterpret_cast<const W*>(base + (uint32_t)(first + ii * step) * row_bytes); a[j] = NT ? __builtin_nontemporal_load(p) : *p; }
#pragma unroll
        for (int j = 0; j < D; ++j) { int ii = i + D + j; if (ii > cnt - 1) ii = cnt - 1; const W* p = reinterpret_cast<const W*>(base + (uint32_t)(first + ii * step) * row_bytes); b[j] = NT ? __builtin_nontemporal_load(p) : *p; }
#pragma unroll
        for (int j = 0; j < D; ++j) acc ^= fold(a[j]);
#pragma unroll
        for (int j = 0; j < D; ++j) acc ^= fold(b[j]);
      }
    }
  }
  if (acc == 0x12345678u) y[0] = 1.f;
}

template <int PATTERN, int T, int D>
__global__ void __launch_bounds__(T * 64) k_stream_buf(const uint32_t* __restrict__ q, int rows, int N, int units_per_wg, float* y) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint32_t acc = 0;
  const int n_strips = (N + 63) / 64;
  const unsigned total = (unsigned)n_strips * rows;
  unsigned g = blockIdx.x * (unsigned)units_per_wg, g_end = g + units_per_wg;
  if (g_end > total) g_end = total;
  const uint32_t row_bytes = 4u * N;
  while (g < g_end) {
    const unsigned strip = g / rows;
    const int r0 = g - strip * rows;
    int r1 = rows;
    if ((unsigned)(r1 - r0) > g_end - g) r1 = r0 + (g_end - g);
    g += r1 - r0;
    int c = strip * 64 + lane;
    if (c > N - 1) c = N - 1;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(q), 0, (uint32_t)r1 * row_bytes, 0x00020000);
    const int n = r1 - r0;
    const int cnt = n > w ? (n - w + T - 1) / T : 0;
    uint32_t voff = 4u * c + (uint32_t)(r0 + w) * row_bytes;
    const uint32_t step = (uint32_t)T * row_bytes;
    if (PATTERN == 4) {
      for (int i = 0; i < cnt; i += 2 * D) {
        uint32_t a[D], b[D];
#pragma unroll
        for (int j = 0; j < D; ++j) { a[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, 0, 2); voff += step; }
#pragma unroll
        for (int j = 0; j < D; ++j) { b[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, 0, 2); voff += step; }
#pragma unroll
        for (int j = 0; j < D; ++j) acc ^= a[j];
#pragma unroll
        for (int j = 0; j < D; ++j) acc ^= b[j];
      }
    } else {
      uint32_t ring[D];
#pragma unroll
      for (int j = 0; j < D; ++j) { ring[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, 0, 2); voff += step; }
      for (int i = 0; i < cnt; i += D) {
#pragma unroll
        for (int j = 0; j < D; j += 2) {
          acc ^= ring[j] ^ ring[j + 1];
          __builtin_amdgcn_sched_barrier(0);
          ring[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, 0, 2); voff += step;
          ring[j + 1] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, 0, 2); voff += step;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  if (acc == 0x12345678u) y[0] = 1.f;
}

struct Shape { int rows, N; const char* name; };

template <int PATTERN, int T, int D, bool NT>
static void run(const Shape& sh, const std::vector<uint32_t*>& qs, float* y, int wgs) {
  constexpr int VEC = (PATTERN == 0 || PATTERN == 3) ? 1 : 4;
  size_t total;
  if (PATTERN == 2) total = (size_t)sh.rows * sh.N / 256;
  else total = (size_t)((sh.N + 64 * VEC - 1) / (64 * VEC)) * sh.rows;
  const int upw = (int)((total + wgs - 1) / wgs);
  const int grid = (int)((total + upw - 1) / upw);
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  auto launch_all = [&]() {
    for (size_t c = 0; c < qs.size(); ++c)
      hipLaunchKernelGGL((k_stream<PATTERN, T, D, NT>), dim3(grid), dim3(T * 64), 0, s, qs[c], sh.rows, sh.N, upw, y);
  };
  launch_all();
  CHECK(hipStreamSynchronize(s));
  hipGraph_t graph; hipGraphExec_t exec;
  CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  launch_all();
  CHECK(hipStreamEndCapture(s, &graph));
  CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  CHECK(hipGraphLaunch(exec, s));
  CHECK(hipStreamSynchronize(s));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  double sum = 0; const int reps = 4;
  for (int r = 0; r < reps; ++r) {
    CHECK(hipEventRecord(e0, s)); CHECK(hipGraphLaunch(exec, s)); CHECK(hipEventRecord(e1, s)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    sum += ms * 1e3 / qs.size();
  }
  const double bytes = (double)sh.rows * sh.N * 4;
  printf("%-8s %5.1f MB pattern %d T=%2d D=%2d %s wgs=%4d grid=%4d: %7.2f us/launch  %6.0f GB/s\n", sh.name, bytes / 1e6, PATTERN, T, D, NT ? "nt" : "  ", wgs, grid, sum / reps, bytes / (sum / reps) / 1e3);
  CHECK(hipGraphExecDestroy(exec)); CHECK(hipGraphDestroy(graph)); CHECK(hipStreamDestroy(s));
}

template <int PATTERN, int T, int D>
static void run_buf(const Shape& sh, const std::vector<uint32_t*>& qs, float* y, int wgs) {
  const size_t total = (size_t)((sh.N + 63) / 64) * sh.rows;
  const int upw = (int)((total + wgs - 1) / wgs);
  const int grid = (int)((total + upw - 1) / upw);
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  auto launch_all = [&]() {
    for (size_t c = 0; c < qs.size(); ++c)
      hipLaunchKernelGGL((k_stream_buf<PATTERN, T, D>), dim3(grid), dim3(T * 64), 0, s, qs[c], sh.rows, sh.N, upw, y);
  };
  launch_all();
  CHECK(hipStreamSynchronize(s));
  hipGraph_t graph; hipGraphExec_t exec;
  CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  launch_all();
  CHECK(hipStreamEndCapture(s, &graph));
  CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  CHECK(hipGraphLaunch(exec, s));
  CHECK(hipStreamSynchronize(s));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  double sum = 0; const int reps = 4;
  for (int r = 0; r < reps; ++r) {
    CHECK(hipEventRecord(e0, s)); CHECK(hipGraphLaunch(exec, s)); CHECK(hipEventRecord(e1, s)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    sum += ms * 1e3 / qs.size();
  }
  const double bytes = (double)sh.rows * sh.N * 4;
  printf("%-8s %5.1f MB pattern %d T=%2d D=%2d nt wgs=%4d grid=%4d: %7.2f us/launch  %6.0f GB/s\n", sh.name, bytes / 1e6, PATTERN, T, D, wgs, grid, sum / reps, bytes / (sum / reps) / 1e3);
  CHECK(hipGraphExecDestroy(exec)); CHECK(hipGraphDestroy(graph)); CHECK(hipStreamDestroy(s));
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  float* y; CHECK(hipMalloc(&y, 16));
  const Shape shapes[] = {{512, 4096, "o_proj"}, {512, 12288, "qkv"}, {512, 22016, "gate+up"}, {1376, 4096, "down"}, {512, 4160, "o_pad"}};
  for (const Shape& sh : shapes) {
    const size_t bytes = (size_t)sh.rows * sh.N * 4;
    int copies = (int)(600e6 / bytes); if (copies > 64) copies = 64; if (copies < 4) copies = 4;
    std::vector<uint32_t*> qs(copies);
    for (auto& p : qs) { CHECK(hipMalloc(&p, bytes)); CHECK(hipMemset(p, 0x5a, bytes)); }
    run<0, 16, 4, true>(sh, qs, y, cus);
    run<0, 16, 8, true>(sh, qs, y, cus);
    run<0, 16, 8, false>(sh, qs, y, cus);
    run<0, 8, 8, true>(sh, qs, y, cus);
    run<0, 8, 16, true>(sh, qs, y, cus);
    run<0, 8, 8, true>(sh, qs, y, 2 * cus);
    run<0, 8, 8, true>(sh, qs, y, 4 * cus);
    run<0, 4, 8, true>(sh, qs, y, 8 * cus);
    run<3, 16, 8, true>(sh, qs, y, cus);
    run_buf<4, 8, 8>(sh, qs, y, cus);
    run_buf<5, 8, 8>(sh, qs, y, cus);
    run_buf<5, 8, 16>(sh, qs, y, cus);
    run_buf<5, 16, 8>(sh, qs, y, cus);
    run<1, 16, 2, true>(sh, qs, y, cus);
    run<1, 16, 4, true>(sh, qs, y, cus);
    run<1, 8, 4, true>(sh, qs, y, cus);
    run<1, 8, 4, true>(sh, qs, y, 2 * cus);
    run<1, 8, 4, true>(sh, qs, y, 4 * cus);
    run<2, 16, 2, true>(sh, qs, y, cus);
    run<2, 16, 4, true>(sh, qs, y, cus);
    run<2, 8, 4, true>(sh, qs, y, 2 * cus);
    run<2, 8, 4, true>(sh, qs, y, 4 * cus);
    run<2, 4, 4, true>(sh, qs, y, 8 * cus);
    run<2, 4, 4, false>(sh, qs, y, 8 * cus);
    for (auto p : qs) CHECK(hipFree(p));
  }
  return 0;
}
