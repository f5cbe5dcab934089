// proto_v2.hip -- stand-alone prototype of the round-2 dense kernel ("v2"), used to settle its
// design on the GPU before it replaced the dense role of squeezellm_amd/csrc/sqllm_kernels.hip:
//   * lane = output column (64-column tile per workgroup), so every k is wave-uniform and vec comes
//     from SGPRs (s_load) -- no DPP, no LDS traffic for x;
//   * codebooks staged as PAIR tables read with ds_read_b64: 4-bit: the BYTE of the packed word
//     (two consecutive k's of one column) addresses a 256-entry table of (lut[lo], lut[hi]) pairs,
//     one v_perm_b32 builds the LDS address; 3-bit: 64-entry pair tables as before;
//     layout [entry][column slot] with the two half-waves in separate banks -> conflict-free;
//   * ONE workgroup per CU, each takes a contiguous range of the flattened (column tile, qweight
//     row) space, so the launch is balanced to the byte whatever the shape.
//
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/proto_v2.hip -o /tmp/proto_v2 && /tmp/proto_v2
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <type_traits>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) float* cfloatp;  // constant address space: scalar loads

__device__ __forceinline__ f32x2 lds_pair(uint32_t a) {
  return *reinterpret_cast<const f32x2 __attribute__((address_space(3)))*>(a);
}

template <int BITS> struct Fmt;
template <> struct Fmt<4> { static constexpr int L = 16, R = 1, KU = 8, ESTRIDE = 256, HALF = 65536, TABLE = 131072; };
template <> struct Fmt<3> { static constexpr int L = 8, R = 3, KU = 32, ESTRIDE = 512, HALF = 0, TABLE = 32768; };

// pair m of a 3-bit unit = bits [6m, 6m+6) of the 96-bit stream (t0, t1, t2); returns field << 9
template <int M>
__device__ __forceinline__ uint32_t field6_x512(uint32_t t0, uint32_t t1, uint32_t t2) {
  constexpr int bit = 6 * M, wd = bit >> 5, o = bit & 31;
  const uint32_t lo = wd == 0 ? t0 : wd == 1 ? t1 : t2;
  uint32_t f;
  if constexpr (o <= 26) {
    if constexpr (o > 9) f = lo >> (o - 9);
    else if constexpr (o < 9) f = lo << (9 - o);
    else f = lo;
  } else {
    const uint32_t hi = wd == 0 ? t1 : t2;
    f = __builtin_amdgcn_alignbit(hi, lo, o) << 9;
  }
  return f & 0x7E00u;
}

__device__ const float kZeros[32] = {0.f};

// MODE bits: 1 = no decode (pure stream: xor the words), 2 = no table build, 4 = no epilogue atomics
// T = waves per workgroup, D = units (4-bit: qweight rows, 3-bit: three-row groups) per chunk, NB =
// chunks a wave keeps in flight (a ring of NB register sets).  PAIR (4-bit only; 3-bit always pairs):
// pair table + ds_read_b64, or the plain 16-entry table + one ds_read_b32 per weight.
//
// Every weight load is an UNCONDITIONAL buffer load; a chunk (or a row of the wave's last, ragged
// chunk) that does not exist gets an offset beyond the descriptor's range: the hardware returns 0
// without touching memory.  So the loop has one static shape -- decode chunk c, refill its register
// set with chunk c + NB -- with exact vmcnt counts, no wasted traffic and no branch around a load.
template <int BITS, int T, int D, int NB, int MODE, bool PAIR = true, int SUBQ = 4>
__global__ void __launch_bounds__(T * 64, 1)
dense_v2(const float* __restrict__ x, const uint32_t* __restrict__ q, const float* __restrict__ lut,
         float* __restrict__ y, int K, int N, int n_tiles, int units_per_wg) {
  using F = Fmt<BITS>;
  constexpr int L = F::L, R = F::R;
  static_assert(PAIR || BITS == 4, "3-bit always uses pair tables");
  constexpr int NENT = PAIR ? L * L : L;         // table entries per column
  constexpr int EPW = NENT / T;                  // table entries a wave builds
  constexpr int NI0 = EPW < L ? EPW : L;         // consecutive first indices
  constexpr int NI1 = EPW / L > 0 ? EPW / L : 1; // second indices
  constexpr int SUB = BITS == 4 ? SUBQ : 1;      // units per decode stage (4-bit: SUBQ qweight rows; 3-bit: one unit)
  static_assert(D % SUB == 0, "whole stages");
  constexpr int TABLE = PAIR ? F::TABLE : 16 * 256;
  __shared__ __attribute__((aligned(16))) char lds[TABLE + 64 * 4 + 16];
  float* red = reinterpret_cast<float*>(lds + TABLE);
  unsigned* ticket = reinterpret_cast<unsigned*>(lds + TABLE + 64 * 4);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned units_total = (unsigned)K / F::KU;
  const unsigned total = (unsigned)n_tiles * units_total;
  unsigned g = blockIdx.x * (unsigned)units_per_wg;
  unsigned g_end = g + (unsigned)units_per_wg;
  if (g_end > total) g_end = total;
  // LDS byte address of this lane's slot in entry row 0
  const uint32_t lane_base = !PAIR ? lane * 4u : BITS == 4 ? ((lane & 31) * 8u) | ((uint32_t)(lane >> 5) << 16) : lane * 8u;
  cfloatp xc = reinterpret_cast<cfloatp>(reinterpret_cast<uintptr_t>(x));
  cfloatp zc = reinterpret_cast<cfloatp>(reinterpret_cast<uintptr_t>(&kZeros[0]));
  const uint32_t row_bytes = 4u * (uint32_t)N;
  // scalar offsets of the D x R loads of a chunk relative to its first row (loop-invariant SGPRs): the
  // per-lane part of an address is ONE VGPR per register set, advanced once per refill
  uint32_t soff[D][R];
#pragma unroll
  for (int j = 0; j < D; ++j)
#pragma unroll
    for (int r = 0; r < R; ++r) soff[j][r] = (uint32_t)(j * T * R + r) * row_bytes;
  if (tid < 64) red[tid] = 0.f;
  if (tid == 64) *ticket = 0u;
  bool first = true;
  while (g < g_end) {
    const unsigned tile = g / units_total;
    const int u0 = (int)(g - tile * units_total);
    int u1 = (int)units_total;
    if ((unsigned)(u1 - u0) > g_end - g) u1 = u0 + (int)(g_end - g);
    g += (unsigned)(u1 - u0);
    const int col = (int)tile * 64 + lane;
    const int colc = col < N ? col : N - 1;
    // ---- loads: codebook values for the table rows this wave builds, then the first NB chunks ----
    const float* lp = lut + (size_t)colc * L;
    float lo_v[NI0], hi_v[NI1];
    const int e0 = w * EPW;                       // first entry this wave builds: i0 = e0 % L, i1 = e0 / L
    if constexpr (!(MODE & 2)) {
      if constexpr (NI0 >= 4) {
#pragma unroll
        for (int i = 0; i < NI0 / 4; ++i) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(lp + (e0 % L) + 4 * i);
          lo_v[4 * i] = t.x; lo_v[4 * i + 1] = t.y; lo_v[4 * i + 2] = t.z; lo_v[4 * i + 3] = t.w;
        }
      } else {
#pragma unroll
        for (int i = 0; i < NI0; ++i) lo_v[i] = lp[(e0 % L) + i];
      }
      if constexpr (PAIR) {
#pragma unroll
        for (int i = 0; i < NI1; ++i) hi_v[i] = lp[e0 / L + i];
      }
    }
    const int n_units = u1 - u0;
    const int n_w = n_units > w ? (n_units - w + T - 1) / T : 0;  // units this wave decodes: u0 + w + T * i
    const int nc = (n_w + D - 1) / D;                              // chunks
    // The descriptor ends with this piece's last row: a unit past it (u0 + w + T * i >= u1, i.e.
    // i >= n_w) is out of range -> the load returns 0 and touches no memory.
    const __amdgpu_buffer_rsrc_t qrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(q), 0, (uint32_t)(u1 * R) * row_bytes, 0x00020000);
    uint32_t voff[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) voff[k] = 4u * (uint32_t)colc + (uint32_t)((u0 + w + k * D * T) * R) * row_bytes;
    auto load_chunk = [&](int k, uint32_t (&dst)[D][R]) {
#pragma unroll
      for (int j = 0; j < D; ++j)
#pragma unroll
        for (int r = 0; r < R; ++r)
          dst[j][r] = __builtin_amdgcn_raw_buffer_load_b32(qrsrc, voff[k], soff[j][r], 2 /* nt */);
      voff[k] += (uint32_t)(NB * D * T * R) * row_bytes;
    };
    uint32_t wbuf[NB][D][R];
#pragma unroll
    for (int k = 0; k < NB; ++k) load_chunk(k, wbuf[k]);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (!first) __syncthreads();  // everybody is done with the previous table
    first = false;
    if constexpr (!(MODE & 2)) {
      if constexpr (PAIR) {
        char* dst = lds + (BITS == 4 ? ((lane & 31) * 8 + (lane >> 5) * F::HALF) : lane * 8) + e0 * F::ESTRIDE;
#pragma unroll
        for (int i1 = 0; i1 < NI1; ++i1)
#pragma unroll
          for (int i0 = 0; i0 < NI0; ++i0)
            *reinterpret_cast<f32x2*>(dst + (i1 * L + i0) * F::ESTRIDE) = f32x2{lo_v[i0], hi_v[i1]};
      } else {
#pragma unroll
        for (int i0 = 0; i0 < NI0; ++i0) *reinterpret_cast<float*>(lds + (e0 + i0) * 256 + lane * 4) = lo_v[i0];
      }
    }
    __syncthreads();

    f32x2 acc[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
    uint32_t xacc = 0;
    // ---- decode: a software pipeline over STAGES (SUB units = NP pair lookups each).  While the
    // packed FMAs of stage s run, the lookups of stage s+1 and its x (scalar loads -> SGPRs) are in
    // flight; scalar loads return out of order, so every wait on this counter is lgkmcnt(0) anyway,
    // and the order  [wait] [issue s+1] [FMAs s]  makes that the natural one.  Stages alternate
    // between two register sets (A, B); NS is even, so a chunk always starts on A.
    constexpr int NP = BITS == 4 ? 4 * SUB : 16;   // pair lookups per stage
    constexpr int NS = D / SUB;                    // stages per chunk
    static_assert(NS % 2 == 0, "stages alternate between two register sets");
    struct St { f32x2 v[NP]; float x[2 * NP]; };
    // issue stage `st` of chunk cc (held in `buf`): lookups + x loads
    auto issue = [&](const uint32_t (&buf)[D][R], int st, int cc, St& o) {
      if constexpr (MODE & 1) {
#pragma unroll
        for (int s2 = 0; s2 < SUB; ++s2)
#pragma unroll
          for (int r = 0; r < R; ++r) xacc ^= buf[st * SUB + s2][r];
        return;
      }
      const uint32_t (*t)[R] = &buf[st * SUB];
      if constexpr (BITS == 4 && PAIR) {
#pragma unroll
        for (int s2 = 0; s2 < SUB; ++s2) {
          o.v[4 * s2 + 0] = lds_pair(__builtin_amdgcn_perm(t[s2][0], lane_base, 0x0C020400u));
          o.v[4 * s2 + 1] = lds_pair(__builtin_amdgcn_perm(t[s2][0], lane_base, 0x0C020500u));
          o.v[4 * s2 + 2] = lds_pair(__builtin_amdgcn_perm(t[s2][0], lane_base, 0x0C020600u));
          o.v[4 * s2 + 3] = lds_pair(__builtin_amdgcn_perm(t[s2][0], lane_base, 0x0C020700u));
        }
      } else if constexpr (BITS == 4) {
#pragma unroll
        for (int s2 = 0; s2 < SUB; ++s2) {
          const uint32_t lo = t[s2][0] & 0x0F0F0F0Fu, hi = (t[s2][0] >> 4) & 0x0F0F0F0Fu;
#define SQ_L(WORD, SEL) *reinterpret_cast<const float __attribute__((address_space(3)))*>(__builtin_amdgcn_perm(WORD, lane_base, SEL))
          o.v[4 * s2 + 0] = f32x2{SQ_L(lo, 0x0C0C0400u), SQ_L(hi, 0x0C0C0400u)};
          o.v[4 * s2 + 1] = f32x2{SQ_L(lo, 0x0C0C0500u), SQ_L(hi, 0x0C0C0500u)};
          o.v[4 * s2 + 2] = f32x2{SQ_L(lo, 0x0C0C0600u), SQ_L(hi, 0x0C0C0600u)};
          o.v[4 * s2 + 3] = f32x2{SQ_L(lo, 0x0C0C0700u), SQ_L(hi, 0x0C0C0700u)};
#undef SQ_L
        }
      } else {
#define SQ_F(M) o.v[M] = lds_pair(lane_base | field6_x512<M>(t[0][0], t[0][1], t[0][2]))
        SQ_F(0); SQ_F(1); SQ_F(2); SQ_F(3); SQ_F(4); SQ_F(5); SQ_F(6); SQ_F(7);
        SQ_F(8); SQ_F(9); SQ_F(10); SQ_F(11); SQ_F(12); SQ_F(13); SQ_F(14); SQ_F(15);
#undef SQ_F
      }
      // x of the stage: units past the wave's last one read zeros (their weight words are zeros too)
#pragma unroll
      for (int s2 = 0; s2 < SUB; ++s2) {
        const int i = cc * D + st * SUB + s2;
        cfloatp px = (i < n_w) ? xc + (size_t)(u0 + w + i * T) * F::KU : zc;
#pragma unroll
        for (int k = 0; k < F::KU; ++k) o.x[s2 * F::KU + k] = px[k];
      }
    };
    auto fmas = [&](const St& o) {
      if constexpr (MODE & 1) return;
#pragma unroll
      for (int m = 0; m < NP; ++m) acc[m & 3] = __builtin_elementwise_fma(o.v[m], f32x2{o.x[2 * m], o.x[2 * m + 1]}, acc[m & 3]);
    };
    St sa, sb;
    issue(wbuf[0], 0, 0, sa);
    __builtin_amdgcn_sched_barrier(0);
    for (int c = 0; c < nc; c += NB) {
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        // (a chunk past the wave's last one decodes zeros against zeros: no branch in this loop)
#pragma unroll
        for (int st = 0; st < NS; st += 2) {
          // the wait for stage s's operands goes BEFORE the issue of stage s+1 (the compiler would
          // put it in front of the FMAs, i.e. behind the issue, and wait for the new lookups too)
          __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
          issue(wbuf[k], st + 1, c + k, sb);
          __builtin_amdgcn_sched_barrier(0);
          fmas(sa);
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_waitcnt(0xC07F);
          if (st + 2 < NS) issue(wbuf[k], st + 2, c + k, sa);
          else issue(wbuf[(k + 1) % NB], 0, c + k + 1, sa);
          __builtin_amdgcn_sched_barrier(0);
          fmas(sb);
          __builtin_amdgcn_sched_barrier(0);
        }
        load_chunk(k, wbuf[k]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    f32x2 a = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    if constexpr (MODE & 1) a.x = __builtin_bit_cast(float, xacc);
    if constexpr (MODE & 4) {
      if (a.x + a.y == 12345.678f) y[0] = 1.f;
    } else {
      // combine without a barrier: LDS float add per wave, the wave drawing the last ticket flushes
      __hip_atomic_fetch_add(red + lane, a.x + a.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      unsigned t = 0;
      if (lane == 0) t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      t = __builtin_amdgcn_readfirstlane(t);
      if (t == T - 1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const float s = red[lane];
        red[lane] = 0.f;
        if (lane == 0) *ticket = 0u;
        if (col < N) atomicAdd(y + col, s);
      }
    }
  }
}

// reference: one thread per column
template <int BITS>
__global__ void ref_kernel(const float* x, const uint32_t* q, const float* lut, float* y, int K, int N) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  constexpr int L = 1 << BITS;
  double s = 0;
  if (BITS == 4) {
    for (int r = 0; r < K / 8; ++r) {
      const uint32_t wv = q[(size_t)r * N + c];
      for (int j = 0; j < 8; ++j) s += (double)lut[(size_t)c * L + ((wv >> (4 * j)) & 15)] * x[8 * r + j];
    }
  } else {
    for (int u = 0; u < K / 32; ++u) {
      const uint32_t t0 = q[(size_t)(3 * u) * N + c], t1 = q[(size_t)(3 * u + 1) * N + c], t2 = q[(size_t)(3 * u + 2) * N + c];
      for (int j = 0; j < 32; ++j) {
        const int bit = 3 * j;
        unsigned long long lo = bit < 64 ? (((unsigned long long)t1 << 32) | t0) >> bit : 0;
        unsigned idx;
        if (bit < 32) idx = (unsigned)((((unsigned long long)t1 << 32) | t0) >> bit) & 7;
        else idx = (unsigned)((((unsigned long long)t2 << 32) | t1) >> (bit - 32)) & 7;
        (void)lo;
        s += (double)lut[(size_t)c * L + idx] * x[32 * u + j];
      }
    }
  }
  y[c] = (float)s;
}

__global__ void touch_kernel(const f32x4* p, size_t n16, float* y) {  // pull a buffer through L2 / MALL
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  f32x4 a = {0, 0, 0, 0};
  for (; i < n16; i += stride) a += p[i];
  if (a.x + a.y + a.z + a.w == 1234.5f) y[0] = 1.f;
}

struct Shape { int K, N; const char* name; };

template <int BITS, int T, int D, int NB, int MODE, bool PAIR = true, int SUBQ = 4>
static double time_variant(const char* label, const Shape& sh, const std::vector<uint32_t*>& qs, const std::vector<float*>& luts,
                           const float* x, float* y, int wgs, bool same_copy, int reps) {
  const int n_tiles = (sh.N + 63) / 64;
  const int units_total = sh.K / Fmt<BITS>::KU;
  const long total = (long)n_tiles * units_total;
  int upw = (int)((total + wgs - 1) / wgs);
  const int grid = (int)((total + upw - 1) / upw);
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  auto launch_all = [&]() {
    for (size_t c = 0; c < qs.size(); ++c) {
      const size_t cc = same_copy ? 0 : c;
      hipLaunchKernelGGL((dense_v2<BITS, T, D, NB, MODE, PAIR, SUBQ>), dim3(grid), dim3(T * 64), 0, s, x, qs[cc], luts[cc], y, sh.K, sh.N, n_tiles, upw);
    }
  };
  launch_all();
  CHECK(hipStreamSynchronize(s));
  hipGraph_t graph;
  hipGraphExec_t exec;
  CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  launch_all();
  CHECK(hipStreamEndCapture(s, &graph));
  CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  CHECK(hipGraphLaunch(exec, s));
  CHECK(hipStreamSynchronize(s));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  double best = 1e30, sum = 0;
  for (int r = 0; r < reps; ++r) {
    CHECK(hipEventRecord(e0, s));
    CHECK(hipGraphLaunch(exec, s));
    CHECK(hipEventRecord(e1, s));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / qs.size();
    sum += us;
    if (us < best) best = us;
  }
  const double bytes = (double)sh.K * sh.N * BITS / 8 + (double)sh.N * (1 << BITS) * 4 + sh.K * 4.0 + 2.0 * sh.N * 4;
  printf("%-10s w%d %-26s %s T=%2d D=%2d SUB=%d NB=%d grid=%4d upw=%5d %s: %7.2f us/launch (best %7.2f)  %6.0f GB/s  frac %.3f\n", sh.name, BITS, label, PAIR ? "pair  " : "direct", T, D, SUBQ, NB, grid, upw,
         same_copy ? "same-copy(MALL)" : "rotating(HBM)  ", sum / reps, best, bytes / (sum / reps) / 1e3, bytes / (sum / reps) / 1e3 / 8000);
  CHECK(hipGraphExecDestroy(exec));
  CHECK(hipGraphDestroy(graph));
  CHECK(hipStreamDestroy(s));
  return sum / reps;
}

template <int BITS>
static void run_shape(const Shape& sh, int cus, bool quick) {
  const size_t qwords = (size_t)sh.K / 32 * BITS * sh.N;
  const size_t qbytes = qwords * 4;
  const int L = 1 << BITS;
  int copies = (int)(600e6 / qbytes);
  if (copies < 4) copies = 4;
  if (copies > 64) copies = 64;
  std::mt19937 rng(1234);
  std::vector<uint32_t> hq(qwords);
  for (auto& v : hq) v = rng();
  std::vector<float> hl((size_t)sh.N * L), hx(sh.K);
  std::normal_distribution<float> nd(0.f, 1.f);
  for (auto& v : hl) v = 0.02f * nd(rng);
  for (auto& v : hx) v = nd(rng);
  std::vector<uint32_t*> qs(copies);
  std::vector<float*> luts(copies);
  for (int c = 0; c < copies; ++c) {
    CHECK(hipMalloc(&qs[c], qbytes));
    CHECK(hipMalloc(&luts[c], hl.size() * 4));
    if (c == 0) {
      CHECK(hipMemcpy(qs[c], hq.data(), qbytes, hipMemcpyHostToDevice));
      CHECK(hipMemcpy(luts[c], hl.data(), hl.size() * 4, hipMemcpyHostToDevice));
    } else {
      CHECK(hipMemcpy(qs[c], qs[0], qbytes, hipMemcpyDeviceToDevice));  // same values, distinct addresses
      CHECK(hipMemcpy(luts[c], luts[0], hl.size() * 4, hipMemcpyDeviceToDevice));
    }
  }
  float *x, *y, *yref;
  CHECK(hipMalloc(&x, sh.K * 4));
  CHECK(hipMalloc(&y, sh.N * 4));
  CHECK(hipMalloc(&yref, sh.N * 4));
  CHECK(hipMemcpy(x, hx.data(), sh.K * 4, hipMemcpyHostToDevice));
  // ---- correctness of every timed variant against the one-thread-per-column reference ----
  hipLaunchKernelGGL(ref_kernel<BITS>, dim3((sh.N + 63) / 64), dim3(64), 0, 0, x, qs[0], luts[0], yref, sh.K, sh.N);
  std::vector<float> hy(sh.N), hr(sh.N);
  CHECK(hipMemcpy(hr.data(), yref, sh.N * 4, hipMemcpyDeviceToHost));
  auto check = [&](auto kern, int T, const char* what, int wgs) {
    const int n_tiles = (sh.N + 63) / 64;
    const long total = (long)n_tiles * (sh.K / Fmt<BITS>::KU);
    const int upw = (int)((total + wgs - 1) / wgs);
    const int grid = (int)((total + upw - 1) / upw);
    CHECK(hipMemset(y, 0, sh.N * 4));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(T * 64), 0, 0, x, qs[0], luts[0], y, sh.K, sh.N, n_tiles, upw);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(hy.data(), y, sh.N * 4, hipMemcpyDeviceToHost));
    double maxd = 0, maxr = 0;
    for (int i = 0; i < sh.N; ++i) { maxd = fmax(maxd, fabs((double)hy[i] - hr[i])); maxr = fmax(maxr, fabs((double)hr[i])); }
    printf("%-10s w%d check %-12s wgs=%d: max|diff|/max|ref| = %.2e %s\n", sh.name, BITS, what, wgs, maxd / maxr, maxd / maxr < 2e-5 ? "OK" : "MISMATCH");
  };
  // D = units per chunk; a 3-bit unit is three dwords, so halve it there
  constexpr int S = BITS == 4 ? 1 : 2;
  check(dense_v2<BITS, 16, 8 / S, 2, 0>, 16, "T16 D8 NB2", cus);
  check(dense_v2<BITS, 16, 4 / S, 3, 0, true, 2>, 16, "T16 D4 NB3", cus - 3);
  check(dense_v2<BITS, 8, 8 / S, 4, 0>, 8, "T8 D8 NB4", cus);
  check(dense_v2<BITS, 8, 8 / S, 3, 0, true, 2>, 8, "T8 D8 NB3", 2 * cus + 1);
  check(dense_v2<BITS, 4, 8 / S, 2, 0>, 4, "T4 D8 NB2", 5);
  if constexpr (BITS == 4) {
    check(dense_v2<4, 8, 8, 2, 0, false>, 8, "direct T8 D8", cus);
    check(dense_v2<4, 16, 4, 4, 0, false, 2>, 16, "direct T16 D4", cus + 7);
  }
  const int reps = 5;
  time_variant<BITS, 8, 8 / S, 2, 0>("full", sh, qs, luts, x, y, cus, false, reps);
  time_variant<BITS, 8, 8 / S, 3, 0>("full", sh, qs, luts, x, y, cus, false, reps);
  time_variant<BITS, 8, 8 / S, 2, 0, true, 2>("full", sh, qs, luts, x, y, cus, false, reps);
  time_variant<BITS, 8, 4 / S, 2, 0, true, 2>("full", sh, qs, luts, x, y, cus, false, reps);
  time_variant<BITS, 8, 4 / S, 4, 0, true, 2>("full", sh, qs, luts, x, y, cus, false, reps);
  time_variant<BITS, 16, 8 / S, 2, 0>("full", sh, qs, luts, x, y, cus, false, reps);
  time_variant<BITS, 16, 4 / S, 2, 0, true, 2>("full", sh, qs, luts, x, y, cus, false, reps);
  time_variant<BITS, 16, 4 / S, 3, 0, true, 2>("full", sh, qs, luts, x, y, cus, false, reps);
  time_variant<BITS, 4, 8 / S, 2, 0>("full", sh, qs, luts, x, y, cus, false, reps);
  if constexpr (BITS == 4) {
    time_variant<4, 8, 8, 2, 0, false>("full", sh, qs, luts, x, y, cus, false, reps);
    time_variant<4, 16, 4, 2, 0, false, 2>("full", sh, qs, luts, x, y, cus, false, reps);
    time_variant<4, 8, 4, 2, 0, false, 2>("full", sh, qs, luts, x, y, cus, false, reps);
  }
  if (!quick) {
    time_variant<BITS, 8, 8 / S, 2, 0>("full", sh, qs, luts, x, y, cus, true, reps);
    time_variant<BITS, 8, 8 / S, 2, 1>("stream only (no decode)", sh, qs, luts, x, y, cus, false, reps);
    time_variant<BITS, 8, 8 / S, 2, 3>("stream, no table, no dec", sh, qs, luts, x, y, cus, false, reps);
    time_variant<BITS, 8, 8 / S, 2, 2>("no table build", sh, qs, luts, x, y, cus, false, reps);
    time_variant<BITS, 8, 8 / S, 2, 4>("no epilogue", sh, qs, luts, x, y, cus, false, reps);
    time_variant<BITS, 8, 8 / S, 2, 7>("nothing but loads", sh, qs, luts, x, y, cus, false, reps);
    time_variant<BITS, 16, 8 / S, 2, 7>("nothing but loads", sh, qs, luts, x, y, cus, false, reps);
    time_variant<BITS, 8, 8 / S, 2, 0>("full, 2 ranges per CU", sh, qs, luts, x, y, 2 * cus, false, reps);
  }
  for (int c = 0; c < copies; ++c) { CHECK(hipFree(qs[c])); CHECK(hipFree(luts[c])); }
  CHECK(hipFree(x)); CHECK(hipFree(y)); CHECK(hipFree(yref));
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs\n", prop.name, cus);
  const bool quick = argc > 1 && argv[1][0] == 'q';
  const Shape shapes[] = {{4096, 4096, "o_proj"}, {4096, 12288, "qkv"}, {4096, 22016, "gate+up"}, {11008, 4096, "down"}};
  for (const Shape& sh : shapes) {
    run_shape<4>(sh, cus, quick);
    run_shape<3>(sh, cus, quick);
  }
  return 0;
}
