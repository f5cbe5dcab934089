// sqllm_kernels.hip -- gfx950 (MI355X / CDNA4) kernels for SqueezeLLM's dense-and-sparse
// LUT-quantised matvec.  Written for wave64 / LDS / HBM3E from scratch; this is not a translation
// of the reference's CUDA kernels (squeezellm/quant_cuda_kernel.cu:741-1164), only their
// arithmetic contract is kept:
//
//   mul[b, n] += sum_k lookup_table[n, idx(k, n)] * vec[b, k]                       (dense, 3/4-bit)
//              + sum_{i in CSR row n} vals[i] * vec[b, cols[i]]                     (outliers)
//              + [n == full_row_indices[c]] sum_k full_rows[k, c] * vec[b, k]       (top-X rows)
//
// ONE launch does all three terms (the reference needs 1-3 dependent launches,
// quant_cuda_kernel.cu:462-504): workgroups are assigned a role by blockIdx.x --
//   [0, csr_blocks)                     CSR chunks, balanced by nnz (not by row)
//   [csr_blocks, +topx_blocks)          top-X row slabs
//   [dense_block0, +dense_blocks)       dense tiles: 256 output columns x one K slice
// and every role accumulates into `mul` with fp32 atomics, as the reference does.
//
// Dense tile design (why it looks the way it does on CDNA4):
//   * qweight is int32 [K/32*bits, N] row-major.  A lane owns 4 adjacent columns and reads them as
//     one 16-byte nontemporal load per qweight row; a wave therefore streams 1 KiB contiguous per
//     row -- full-width coalesced HBM traffic, read exactly once.
//   * the 4 waves of a workgroup share the same 256 columns and split the tile's K slice, so all
//     64 lanes of a wave always work on the SAME k: vec[k] is wave-uniform.  32 k's are fetched by
//     one coalesced dword load and broadcast with v_readlane into the FMA's scalar operand -- one
//     extra op per 4 weights and no LDS traffic (the reference reads vec from shared memory once
//     per weight, quant_cuda_kernel.cu:866).
//   * the per-channel codebooks of the tile live in LDS as 4 sub-tables (one per dword of the
//     lane's 16-byte load) laid out [entry][lane]: a lookup is one ds_read_b32 whose bank is a
//     function of the lane only, so lookups never conflict whatever the indices are.  A per-sub-
//     table lane rotation makes the transposing staging writes conflict-free too.
//   * per-lane partial sums are combined across the 4 waves through LDS and leave the workgroup
//     as one atomic per column (K-slice-way contention instead of the reference's K/128-way).
//   * no MFMA: batch-1 decode is a memory-bound gather; the ceilings are HBM, then LDS lookup
//     issue (one ds_read_b32 per weight), then VALU.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sqllm_kernels.h"

namespace sqllm {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int BITS> struct Fmt;
template <> struct Fmt<4> {
  static constexpr int kLut = 16;   // codebook entries per channel
  static constexpr int kRows = 1;   // qweight rows per group
  static constexpr int kK = 8;      // weights (k's) per group per column
};
template <> struct Fmt<3> {
  static constexpr int kLut = 8;
  static constexpr int kRows = 3;   // 3 rows hold 32 3-bit fields (squeezellm/quant.py:185-203)
  static constexpr int kK = 32;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// ------------------------------------------------------------------------------------------------
// 3-bit field extraction.  The three rows of a group form one little-endian 96-bit stream in which
// weight k occupies bits [3k, 3k+3): row0 bits 0-29 are k0..9, row0[30:31] + row1[0] are k10,
// row1[1:30] are k11..20, row1[31] + row2[0:1] are k21, row2[2:31] are k22..31 -- exactly the
// layout pack2 writes (squeezellm/quant.py:185-203) and the reference decodes with its two
// "straddler" expressions (quant_cuda_kernel.cu:792, :809).
// Returns the field already shifted to bits [8, 11) (i.e. index * 256), ready to be OR-ed into an
// LDS byte address.
// ------------------------------------------------------------------------------------------------
template <int KIDX>
__device__ __forceinline__ uint32_t field3_x256(uint32_t t0, uint32_t t1, uint32_t t2) {
  constexpr int bit = 3 * KIDX;
  constexpr int w = bit >> 5;
  constexpr int o = bit & 31;
  const uint32_t lo = (w == 0) ? t0 : (w == 1) ? t1 : t2;
  uint32_t f;
  if constexpr (o <= 29) {
    if constexpr (o > 8) f = lo >> (o - 8);
    else if constexpr (o < 8) f = lo << (8 - o);
    else f = lo;
  } else {
    const uint32_t hi = (w == 0) ? t1 : t2;
    f = __builtin_amdgcn_alignbit(hi, lo, o) << 8;
  }
  return f & 0x700u;
}

template <int P>
__device__ __forceinline__ uint32_t field4_x256(uint32_t t) {
  uint32_t f;
  if constexpr (4 * P > 8) f = t >> (4 * P - 8);
  else if constexpr (4 * P < 8) f = t << (8 - 4 * P);
  else f = t;
  return f & 0xF00u;
}

// ------------------------------------------------------------------------------------------------
// dense role
//
// Codebook layout in LDS (bytes):  addr(j, idx, lane) = j * SUBB + idx * 256 + 4 * slot(lane, j)
//   j     = which dword of the lane's 16-byte load (the lane's j-th column)
//   SUBB  = 2^BITS * 256, so idx * 256 can be OR-ed into a per-lane base whose bits [8, 8+BITS) are 0
//           (the __shared__ array is the kernel's only LDS object and sits at LDS address 0)
//   slot  = (lane + 8 j) & 63: a per-sub-table lane rotation.  Lookups by the 64 lanes of a wave
//           touch 64 different dwords of one 256-byte row -> conflict-free for any indices; the
//           staging writes of threads 4l..4l+3 (same l, j = 0..3) land 8 slots apart, so a
//           32-thread group writes 32 different banks too.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float lds_read_f32(uint32_t byte_addr) {
  return *reinterpret_cast<const float __attribute__((address_space(3)))*>(byte_addr);
}

__device__ __forceinline__ float bcast_lane(float v, int src_lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src_lane));
}

// vec[k] handling: all 64 lanes of a wave always work on the same k, so x is wave-uniform.  Each
// batch of 32 k's is fetched by ONE coalesced dword load (lane l holds x[k0 + (l & 31)], vmcnt
// domain, prefetched with the weights) and broadcast with v_readlane into the SGPR operand of the
// FMAs: +1 scalar-producing op per k (per 4 weights), no LDS traffic, no lgkmcnt interference
// with the lookups, no SGPR pressure at any batch tile.

// Pin a value: nothing that consumes it can be placed above this point, and the statement is
// ordered against the other pins / scheduling fences.  Used to keep each decode stage's shifts
// from being hoisted to the top of the loop body by instruction selection (which then spills).
#define SQLLM_PIN4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#define SQLLM_PIN3(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c))

// One qweight row of a 4-bit tile: this lane's 4 columns x 8 weights.  Stage 1 issues the 32
// lookups, stage 2 the FMAs; the fence at the end closes the stage.
template <int BT, int LANE0>
__device__ __forceinline__ void row4(u32x4 w, const uint32_t (&tb)[4], const float (&xv)[BT],
                                     float (&acc)[4][BT]) {
  float v[4][8];
  uint32_t t[4] = {w.x, w.y, w.z, w.w};
  SQLLM_PIN4(t[0], t[1], t[2], t[3]);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j][0] = lds_read_f32(tb[j] | field4_x256<0>(t[j]));
    v[j][1] = lds_read_f32(tb[j] | field4_x256<1>(t[j]));
    v[j][2] = lds_read_f32(tb[j] | field4_x256<2>(t[j]));
    v[j][3] = lds_read_f32(tb[j] | field4_x256<3>(t[j]));
    v[j][4] = lds_read_f32(tb[j] | field4_x256<4>(t[j]));
    v[j][5] = lds_read_f32(tb[j] | field4_x256<5>(t[j]));
    v[j][6] = lds_read_f32(tb[j] | field4_x256<6>(t[j]));
    v[j][7] = lds_read_f32(tb[j] | field4_x256<7>(t[j]));
  }
#pragma unroll
  for (int p = 0; p < 8; ++p)
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      const float xs = bcast_lane(xv[b], LANE0 + p);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j][b] = __builtin_fmaf(v[j][p], xs, acc[j][b]);
    }
  __builtin_amdgcn_sched_barrier(0);
}

template <int KI, int N>
struct Lookup3 {
  static __device__ __forceinline__ void run(uint32_t t0, uint32_t t1, uint32_t t2, uint32_t tb, float (&v)[32]) {
    v[KI] = lds_read_f32(tb | field3_x256<KI>(t0, t1, t2));
    if constexpr (KI + 1 < N) Lookup3<KI + 1, N>::run(t0, t1, t2, tb, v);
  }
};

// One column of a 3-bit group: 32 weights from three dwords.
template <int BT>
__device__ __forceinline__ void col3(uint32_t t0, uint32_t t1, uint32_t t2, uint32_t tb,
                                     const float (&xv)[BT], float (&acc)[BT]) {
  float v[32];
  SQLLM_PIN3(t0, t1, t2);
  Lookup3<0, 32>::run(t0, t1, t2, tb, v);
#pragma unroll
  for (int p = 0; p < 32; ++p)
#pragma unroll
    for (int b = 0; b < BT; ++b) acc[b] = __builtin_fmaf(v[p], bcast_lane(xv[b], p), acc[b]);
  __builtin_amdgcn_sched_barrier(0);
}

template <int BITS, int BT>
__device__ __forceinline__ void dense_role(const float* x, const u32x4* q,
                                           float* __restrict__ y, const float* __restrict__ lut,
                                           int K, int N, int b0, int nb, int bid, int n_col_tiles,
                                           int groups_total, int gpw, float* lds) {
  using F = Fmt<BITS>;
  constexpr int L = F::kLut;
  constexpr int SUBB = L * 256;  // bytes per sub-table
  constexpr int R = F::kRows;    // qweight rows per group
  constexpr int U = 32 / F::kK;  // groups per batch: a batch is always 32 k's (4 rows w4, 3 rows w3)
  constexpr int RB = U * R;      // qweight rows per batch
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = bid % n_col_tiles;
  const int ks = bid / n_col_tiles;
  const int col0 = ct * kTileN;

  // ---- this wave's K range, in groups ----
  const int g_beg = (ks * kWaves + wave) * gpw;
  int g_end = g_beg + gpw;
  if (g_end > groups_total) g_end = groups_total;

  // Loads are UNCONDITIONAL with clamped addresses (a conditional load becomes a branch with an
  // immediate vmcnt(0), which kills the prefetch): lanes past N re-read the last valid 16 bytes
  // of the row, rows past the end of the matrix re-read its last row; such data is never used.
  const int row_stride = N / 4;  // in 16-byte units
  int cidx = col0 / 4 + lane;
  if (cidx > row_stride - 1) cidx = row_stride - 1;
  const int last_row = groups_total * R - 1;
  const u32x4* qcol = q + cidx;
  auto load_row = [&](int row) -> u32x4 {
    if (row > last_row) row = last_row;  // scalar min
    return __builtin_nontemporal_load(qcol + (size_t)row * row_stride);
  };

  // x: lane l holds x[b][k_batch + (l & 31)]; a batch is 32 k's
  const float* xrow[BT];
#pragma unroll
  for (int b = 0; b < BT; ++b) xrow[b] = x + (size_t)(b0 + (b < nb ? b : nb - 1)) * K;
  auto load_x = [&](int b, int k_batch) -> float {
    int k = k_batch + (lane & 31);
    if (k > K - 1) k = K - 1;
    return xrow[b][k];
  };

  // ---- first batch of weight loads goes out before anything else ----
  u32x4 cur[RB], nxt[RB];
  float xcur[BT], xnxt[BT];
  int row = g_beg * R;
#pragma unroll
  for (int r = 0; r < RB; ++r) cur[r] = load_row(row + r);
#pragma unroll
  for (int b = 0; b < BT; ++b) xcur[b] = load_x(b, g_beg * F::kK);
  // keep the prologue loads ABOVE the codebook staging and its barrier (LLVM would otherwise sink
  // them into the loop preheader, serialising the first HBM round trip behind the LUT's)
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);

  // ---- stage the tile's codebooks: thread t owns column col0 + t ----
  {
    const int c = col0 + tid;
    const int j = tid & 3, l = tid >> 2;
    const int slot = (l + 8 * j) & 63;
    float* dst = lds + (j * SUBB) / 4 + slot;
    const f32x4* src = reinterpret_cast<const f32x4*>(lut + (size_t)(c < N ? c : N - 1) * L);
    f32x4 e[L / 4];
#pragma unroll
    for (int v4 = 0; v4 < L / 4; ++v4) e[v4] = src[v4];
#pragma unroll
    for (int v4 = 0; v4 < L / 4; ++v4) {
      dst[(4 * v4 + 0) * 64] = e[v4].x;
      dst[(4 * v4 + 1) * 64] = e[v4].y;
      dst[(4 * v4 + 2) * 64] = e[v4].z;
      dst[(4 * v4 + 3) * 64] = e[v4].w;
    }
  }

  float acc[4][BT];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int b = 0; b < BT; ++b) acc[j][b] = 0.f;

  // per-lane LDS byte bases of the four sub-tables (rotation as in staging)
  uint32_t tb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) tb[j] = j * SUBB + 4 * ((lane + 8 * j) & 63);

  __syncthreads();  // codebooks visible

  for (int g = g_beg; g < g_end; g += U) {
    // prefetch the next batch while this one is decoded
    row += RB;
#pragma unroll
    for (int r = 0; r < RB; ++r) nxt[r] = load_row(row + r);
#pragma unroll
    for (int b = 0; b < BT; ++b) xnxt[b] = load_x(b, (g + U) * F::kK);
    __builtin_amdgcn_sched_barrier(0);

    if constexpr (BITS == 4) {
      row4<BT, 0>(cur[0], tb, xcur, acc);
      if (g + 1 < g_end) row4<BT, 8>(cur[1], tb, xcur, acc);   // wave-uniform tails
      if (g + 2 < g_end) row4<BT, 16>(cur[2], tb, xcur, acc);
      if (g + 3 < g_end) row4<BT, 24>(cur[3], tb, xcur, acc);
    } else {
      col3<BT>(cur[0].x, cur[1].x, cur[2].x, tb[0], xcur, acc[0]);
      col3<BT>(cur[0].y, cur[1].y, cur[2].y, tb[1], xcur, acc[1]);
      col3<BT>(cur[0].z, cur[1].z, cur[2].z, tb[2], xcur, acc[2]);
      col3<BT>(cur[0].w, cur[1].w, cur[2].w, tb[3], xcur, acc[3]);
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) cur[r] = nxt[r];
#pragma unroll
    for (int b = 0; b < BT; ++b) xcur[b] = xnxt[b];
  }

  // ---- combine the 4 waves through LDS (codebooks are dead now), one atomic per column ----
  __syncthreads();
  float* red = lds;  // [wave][b][256]
#pragma unroll
  for (int b = 0; b < BT; ++b) {
    f32x4 v = {acc[0][b], acc[1][b], acc[2][b], acc[3][b]};
    *reinterpret_cast<f32x4*>(red + (wave * BT + b) * kTileN + 4 * lane) = v;
  }
  __syncthreads();
  const int c = col0 + tid;
  if (c < N) {
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      if (b < nb) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) s += red[(w * BT + b) * kTileN + tid];
        atomicAdd(y + (size_t)(b0 + b) * N + c, s);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// CSR role: one workgroup per chunk of kCsrChunk consecutive non-zeros (balanced by nnz, so a few
// very long rows cost nothing extra -- the reference walks one row per thread serially,
// quant_cuda_kernel.cu:1049-1058).
// ------------------------------------------------------------------------------------------------

// Largest r in [0, n_entries) with rows[r] <= target, assuming rows is non-decreasing and
// rows[0] <= target.  256-ary cooperative search: each round is one coalesced probe + a count.
__device__ __forceinline__ int coop_last_le(const int* __restrict__ rows, int n_entries, int target) {
  int lo = 0, hi = n_entries;
  while (hi - lo > 1) {
    const int step = (hi - lo + kThreads - 1) / kThreads;
    const int i = lo + (int)threadIdx.x * step;
    const int pred = (i < hi) && (rows[i] <= target);
    const int cnt = __syncthreads_count(pred);
    const int nlo = lo + (cnt > 0 ? cnt - 1 : 0) * step;
    int nhi = nlo + step;
    if (nhi > hi) nhi = hi;
    lo = nlo;
    hi = nhi;
  }
  return lo;
}

__device__ __forceinline__ void csr_role(const float* x, float* __restrict__ y,
                                         const int* __restrict__ rows, const int* __restrict__ cols,
                                         const float* __restrict__ vals, int nnz, int K, int N, int b0,
                                         int nb, int chunk, float* lds) {
  const int tid = threadIdx.x;
  const int e0 = chunk * kCsrChunk;
  int e1 = e0 + kCsrChunk;
  if (e1 > nnz) e1 = nnz;
  if (e0 >= e1) return;

  const int r_lo = coop_last_le(rows, N + 1, e0);
  const int r_hi = coop_last_le(rows, N + 1, e1 - 1);
  const int nrows = r_hi - r_lo + 1;
  const bool in_lds = nrows <= kCsrSpanMax;

  int* srows = reinterpret_cast<int*>(lds);           // [kCsrSpanMax]
  float* sacc = lds + kCsrSpanMax;                    // [kCsrSpanMax]
  if (in_lds)
    for (int i = tid; i < nrows; i += kThreads) srows[i] = rows[r_lo + i];
  __syncthreads();

  // this thread's elements: e0 + tid + 256*i  (coalesced), their local row and operands
  constexpr int EPT = kCsrChunk / kThreads;
  int lr[EPT], col[EPT];
  float val[EPT];
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    const int e = e0 + tid + kThreads * i;
    const bool ok = e < e1;
    col[i] = ok ? cols[e] : 0;
    val[i] = ok ? vals[e] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    const int e = e0 + tid + kThreads * i;
    int lo = 0, hi = nrows;
    if (in_lds) {
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (srows[mid] <= e) lo = mid; else hi = mid;
      }
    } else {
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (rows[r_lo + mid] <= e) lo = mid; else hi = mid;
      }
    }
    lr[i] = (e < e1) ? lo : -1;
  }

  for (int b = 0; b < nb; ++b) {
    const float* xb = x + (size_t)(b0 + b) * K;
    float* yb = y + (size_t)(b0 + b) * N + r_lo;
    if (in_lds) {
      for (int i = tid; i < nrows; i += kThreads) sacc[i] = 0.f;
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const float p = val[i] * xb[col[i]];
      const int r = lr[i];
      // a wave holds 64 consecutive non-zeros: inside a long row they all share the row, so
      // reduce in registers and issue one atomic instead of 64 colliding ones
      const int r_first = __builtin_amdgcn_readfirstlane(r);
      if (__all(r == r_first)) {
        const float s = wave_sum(p);
        if ((tid & 63) == 0 && r_first >= 0) {
          if (in_lds) atomicAdd(sacc + r_first, s); else atomicAdd(yb + r_first, s);
        }
      } else if (r >= 0) {
        if (in_lds) atomicAdd(sacc + r, p); else atomicAdd(yb + r, p);
      }
    }
    if (in_lds) {
      __syncthreads();
      for (int i = tid; i < nrows; i += kThreads) {
        const float s = sacc[i];
        if (s != 0.f) atomicAdd(yb + i, s);
      }
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------------
// top-X role: full_rows is fp32 [K, topX] row-major; a workgroup takes kTopxRows consecutive k's,
// i.e. one contiguous slab of kTopxRows*topX floats, and streams it coalesced (the reference keeps
// topX of 128 lanes busy with stride-topX reads, quant_cuda_kernel.cu:1113-1118).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void topx_role(const float* x, float* __restrict__ y,
                                          const float* __restrict__ full_rows,
                                          const int* __restrict__ full_idx, int topX, int K, int N,
                                          int b0, int nb, int slab, float* lds) {
  const int tid = threadIdx.x;
  const int k0 = slab * kTopxRows;
  int k1 = k0 + kTopxRows;
  if (k1 > K) k1 = K;
  const int nel = (k1 - k0) * topX;
  const float* fr = full_rows + (size_t)k0 * topX;
  const bool in_lds = topX <= kTopxLds;
  float* sacc = lds;
  for (int b = 0; b < nb; ++b) {
    const float* xb = x + (size_t)(b0 + b) * K + k0;
    float* yb = y + (size_t)(b0 + b) * N;
    if (in_lds) {
      for (int c = tid; c < topX; c += kThreads) sacc[c] = 0.f;
      __syncthreads();
    }
    for (int e = tid; e < nel; e += kThreads) {
      const int kk = e / topX;
      const int c = e - kk * topX;
      const float p = fr[e] * xb[kk];
      if (in_lds) atomicAdd(sacc + c, p); else atomicAdd(yb + full_idx[c], p);
    }
    if (in_lds) {
      __syncthreads();
      for (int c = tid; c < topX; c += kThreads) atomicAdd(yb + full_idx[c], sacc[c]);
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------------
// the fused kernel
// ------------------------------------------------------------------------------------------------
template <int BITS, int BT>
__global__ void __launch_bounds__(kThreads, 4)
sqllm_fused_matvec(const float* x, const u32x4* q, float* __restrict__ y,
                   const float* __restrict__ lut, const int* __restrict__ rows,
                   const int* __restrict__ cols, const float* __restrict__ vals,
                   const float* __restrict__ full_rows, const int* __restrict__ full_idx,
                   KernelGeom gm) {
  __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
  const int bid = blockIdx.x;
  const int b0 = blockIdx.y * BT;
  int nb = gm.batch - b0;
  if (nb > BT) nb = BT;

  if (bid >= gm.dense_block0) {
    const int d = bid - gm.dense_block0;
    if (d < gm.dense_blocks)
      dense_role<BITS, BT>(x, q, y, lut, gm.K, gm.N, b0, nb, d, gm.col_tiles, gm.groups_total,
                           gm.groups_per_wave, lds);
  } else if (bid < gm.csr_blocks) {
    csr_role(x, y, rows, cols, vals, gm.nnz, gm.K, gm.N, b0, nb, bid, lds);
  } else if (bid < gm.csr_blocks + gm.topx_blocks) {
    topx_role(x, y, full_rows, full_idx, gm.topX, gm.K, gm.N, b0, nb, bid - gm.csr_blocks, lds);
  }
}

template <int BITS, int BT>
static hipError_t launch_inst(const LaunchArgs& a, hipStream_t stream) {
  dim3 grid(a.gm.dense_block0 + a.gm.dense_blocks, (a.gm.batch + BT - 1) / BT);
  hipLaunchKernelGGL((sqllm_fused_matvec<BITS, BT>), grid, dim3(kThreads), 0, stream, a.x,
                     reinterpret_cast<const u32x4*>(a.q), a.y, a.lut, a.rows, a.cols, a.vals,
                     a.full_rows, a.full_idx, a.gm);
  return hipGetLastError();
}

hipError_t launch_fused(int bits, const LaunchArgs& a, hipStream_t stream) {
  const int bt = batch_tile(a.gm.batch);
  if (bits == 4) {
    switch (bt) {
      case 1: return launch_inst<4, 1>(a, stream);
      case 2: return launch_inst<4, 2>(a, stream);
      case 4: return launch_inst<4, 4>(a, stream);
      default: return launch_inst<4, 8>(a, stream);
    }
  }
  switch (bt) {
    case 1: return launch_inst<3, 1>(a, stream);
    case 2: return launch_inst<3, 2>(a, stream);
    case 4: return launch_inst<3, 4>(a, stream);
    default: return launch_inst<3, 8>(a, stream);
  }
}

}  // namespace sqllm
