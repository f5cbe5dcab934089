// sqllm_pair.hip -- 4-bit batch-1 dense role with COLUMN-PAIR codebook tables, 16-wave workgroups.
//
// Why (round 3, tools/sweep.py --ablate on the fused kernel with the half-stage ablation fixed): with the
// decode compiled out the fused kernel runs the 7B launches in 4.7 / 6.3 / 10.0 / 6.1 us against 5.2 / 8.7 /
// 13.4 / 8.2 us with it -- the decode costs 2.1-3.3 us on the three large launches, and with the LDS lookups
// alone compiled out next to nothing changes: it is the VECTOR instructions of the decode (one v_perm address
// per weight, the nibble split, half a packed FMA, a quarter DPP move: 68 per 32 weights of a lane) that cost,
// not the lookups.  A lookup that returns TWO weights halves the address instructions:
//
//   * the codebooks of an (even, odd) column pair are staged as ONE table of 256 entries
//         entry[ia + 16 * ib] = (lut_even[ia], lut_odd[ib])                       8 bytes
//     and the byte  idx_even(k) | idx_odd(k) << 4  -- built for four k's at a time with two v_bfi from the two
//     columns' words -- addresses ONE ds_read_b64 that returns both columns' weights of row k; one packed FMA
//     multiplies the pair by x[k].  Per 32 weights of a lane: 8 (byte words) + 16 (v_perm) + 16 (v_pk_fma) + 8 (DPP)
//     = 48 vector instructions and 16 lookups instead of 68 and 32.
//   * a 64-column tile has 32 column pairs: 32 x 256 x 8 B = 64 KiB of tables (entry rows of 256 B: the even
//     pairs of the 16 column groups in the low 128 B, the odd ones in the high, so that the address is the same
//     byte permute as in the fused kernel).  Two such workgroups fit a CU; to keep 32 waves per CU (the fused
//     kernels' occupancy is what hides their latencies) a workgroup has SIXTEEN waves, which also halves the
//     table-building cost per weight.
// Everything else is the fused kernel's: same tile shape and lane layout (a lane owns 4 adjacent columns = two
// pairs, 16 lanes cover the tile, a wave load covers 4 rows), x through DPP row broadcasts, partial sums
// through per-wave LDS slabs and a ticket, one atomic per column and workgroup; the sparse roles run in the
// first eight waves of their workgroups.  Reference arithmetic: squeezellm/quant_cuda_kernel.cu:831-880.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "sqllm_decode.h"
#include "sqllm_roles.h"

namespace sqllm {

constexpr int kPairWaves = 16;
constexpr int kPairTableBytes = 65536;
constexpr int kPairLdsBytes = kPairTableBytes + kPairWaves * kTileN * 4 + 16;  // tables, slabs [wave][64], ticket

// Packed FMAs of one column pair against 8 consecutive x values (x broadcast into the halves of four register
// pairs, each FMA picks its half through op_sel -- see sqllm_stream.hip: fma_pair_xp for why not f32x2{x, x}).
// `lo[i]` = the pair's weights at k = 2i, `hi[i]` at k = 2i + 1.
template <int XL>
__device__ __forceinline__ void fma_pairs_interleaved(const f32x2 (&lo)[4], const f32x2 (&hi)[4], float xv, f32x2& acc) {
  const f32x2 x01 = {row_bcast<XL + 0>(xv), row_bcast<XL + 1>(xv)}, x23 = {row_bcast<XL + 2>(xv), row_bcast<XL + 3>(xv)};
  const f32x2 x45 = {row_bcast<XL + 4>(xv), row_bcast<XL + 5>(xv)}, x67 = {row_bcast<XL + 6>(xv), row_bcast<XL + 7>(xv)};
  f32x2 a = acc;
#define SQLLM_PKFMA_LO(V, X) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a) : "v"(V), "v"(X))
#define SQLLM_PKFMA_HI(V, X) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(a) : "v"(V), "v"(X))
  SQLLM_PKFMA_LO(lo[0], x01); SQLLM_PKFMA_HI(hi[0], x01);
  SQLLM_PKFMA_LO(lo[1], x23); SQLLM_PKFMA_HI(hi[1], x23);
  SQLLM_PKFMA_LO(lo[2], x45); SQLLM_PKFMA_HI(hi[2], x45);
  SQLLM_PKFMA_LO(lo[3], x67); SQLLM_PKFMA_HI(hi[3], x67);
#undef SQLLM_PKFMA_LO
#undef SQLLM_PKFMA_HI
  acc = a;
}

// One qweight row of this lane's 4 columns (two pairs) x 8 weights.
//   lo = bytes { idx_a(k) | idx_b(k) << 4 : k = 0, 2, 4, 6 },  hi = the same for k = 1, 3, 5, 7
// address of a lookup = [byte 1 = the byte, byte 0 = 8 * column group] (+ 128 for the lane's second pair).
// Bank conflicts: ds_read_b64 is served 32 lanes (two 16-lane rows) at a time over 64 banks, and a lookup's
// bank is (pair half, column group) -- the two rows of a half-wave would always meet on the same 32 banks with
// different entries (a 2-way conflict on EVERY lookup).  So the odd lane rows take their two column pairs in
// the opposite order: in each instruction the even rows read the low 128 bytes of the entry rows, the odd
// rows the high ones.  `odd` selects (4 v_cndmask per row of 32 weights); acc[0] / acc[1] of an odd row hold
// the pairs swapped and are swapped back once, before the rows are folded.
template <int XL, int ABL>
__device__ __forceinline__ void step4_pair(const u32x4& slot, float xv, bool odd, uint32_t off_first, uint32_t off_second,
                                           f32x2 (&acc)[2]) {
  uint32_t t[4] = {odd ? slot.z : slot.x, odd ? slot.w : slot.y, odd ? slot.x : slot.z, odd ? slot.y : slot.w};
  SQLLM_PIN4(t[0], t[1], t[2], t[3]);
  if constexpr (ABL & 2) {
    acc[0].x += __builtin_bit_cast(float, t[0] ^ t[1]) * xv;
    acc[1].x += __builtin_bit_cast(float, t[2] ^ t[3]) * xv;
    return;
  }
#pragma unroll
  for (int jp = 0; jp < 2; ++jp) {
    const uint32_t a = t[2 * jp], b = t[2 * jp + 1];
    const uint32_t lo = (a & 0x0F0F0F0Fu) | ((b << 4) & 0xF0F0F0F0u);  // v_lshlrev + v_bfi
    const uint32_t hi = ((a >> 4) & 0x0F0F0F0Fu) | (b & 0xF0F0F0F0u);  // v_lshrrev + v_bfi
    const uint32_t lane_off = jp ? off_second : off_first;
    f32x2 wl[4], wh[4];
    wl[0] = lds_read_f32x2(__builtin_amdgcn_perm(lo, lane_off, 0x0C0C0400u));
    wh[0] = lds_read_f32x2(__builtin_amdgcn_perm(hi, lane_off, 0x0C0C0400u));
    wl[1] = lds_read_f32x2(__builtin_amdgcn_perm(lo, lane_off, 0x0C0C0500u));
    wh[1] = lds_read_f32x2(__builtin_amdgcn_perm(hi, lane_off, 0x0C0C0500u));
    wl[2] = lds_read_f32x2(__builtin_amdgcn_perm(lo, lane_off, 0x0C0C0600u));
    wh[2] = lds_read_f32x2(__builtin_amdgcn_perm(hi, lane_off, 0x0C0C0600u));
    wl[3] = lds_read_f32x2(__builtin_amdgcn_perm(lo, lane_off, 0x0C0C0700u));
    wh[3] = lds_read_f32x2(__builtin_amdgcn_perm(hi, lane_off, 0x0C0C0700u));
    fma_pairs_interleaved<XL>(wl, wh, xv, acc[jp]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int ABL>
__device__ __forceinline__ void dense_role_pair4(const float* x, const char* qbase, float* __restrict__ y, const float* lut,
                                                 int K, int N, int bid, int n_col_tiles, int units_total, int units_per_wg,
                                                 char* lds) {
  constexpr int WAVES = kPairWaves;
  constexpr int NBUF = 4;          // steps per chunk (pairs of steps share an x register)
  constexpr int STEP = WAVES * 4;  // rows a workgroup step covers
  __builtin_amdgcn_s_waitcnt(0);   // (clean slate for the compiler's wait-count model, see dense_role)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, grp = lane >> 4;
  const int ct = bid % n_col_tiles;
  const int ks = bid / n_col_tiles;
  const int col0 = ct * kTileN;
  const int u_beg = ks * units_per_wg;
  int u_end = u_beg + units_per_wg;
  if (u_end > units_total) u_end = units_total;
  const int u_last = u_end - 1;
  const int u_wave = u_beg + wave * 4;

  const int row_stride = N / 4;  // in 16-byte units
  int cidx = col0 / 4 + i16;
  if (cidx > row_stride - 1) cidx = row_stride - 1;
  const uint32_t lane_bytes = 16u * (uint32_t)cidx;
  const uint32_t row_bytes = 16u * (uint32_t)row_stride;
  const char* xb = reinterpret_cast<const char*>(x);

  // unconditional loads with clamped addresses (steps past the slice's end re-read its last row: cache hits)
  auto load_chunk = [&](int u, u32x4 (&w)[NBUF], float (&xs)[NBUF / 2]) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < NBUF; ++s) {
      int uu = u + grp + s * STEP;
      if (uu > u_last) uu = u_last;
      w[s] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(qbase + (__umul24((uint32_t)uu, row_bytes) + lane_bytes)));
    }
#pragma unroll
    for (int s2 = 0; s2 < NBUF / 2; ++s2) {  // lanes 0-7 of a row hold the first step's 8 k's, lanes 8-15 the second step's
      int uu = u + grp + (2 * s2 + (i16 >> 3)) * STEP;
      if (uu > u_last) uu = u_last;
      xs[s2] = *reinterpret_cast<const float*>(xb + 4u * (8u * (uint32_t)uu + (i16 & 7)));
    }
  };

  // ---- codebook loads: thread = (column group s = tid & 15, pair jp = (tid >> 4) & 1, ib = (tid >> 5) & 15, half = tid >> 9)
  //      writes the 8 entries ia = 8 * half .. + 7 of its pair's table row block ib ----
  const int st_s = tid & 15, st_jp = (tid >> 4) & 1, st_ib = (tid >> 5) & 15, st_half = tid >> 9;
  float ea[8], eb = 0.f;
  if constexpr (!(ABL & 4)) {
    int ca = col0 + 4 * st_s + 2 * st_jp;
    if (ca > N - 2) ca = N - 2;  // (N % 4 == 0: a clamped pair stays a pair)
    const float* pa = lut + (size_t)ca * 16 + 8 * st_half;
    const f32x4 t0 = *reinterpret_cast<const f32x4*>(pa), t1 = *reinterpret_cast<const f32x4*>(pa + 4);
    ea[0] = t0.x; ea[1] = t0.y; ea[2] = t0.z; ea[3] = t0.w; ea[4] = t1.x; ea[5] = t1.y; ea[6] = t1.z; ea[7] = t1.w;
    eb = lut[(size_t)(ca + 1) * 16 + st_ib];
  }
  float* slabs = reinterpret_cast<float*>(lds + kPairTableBytes);                       // [wave][64]
  unsigned* ticket = reinterpret_cast<unsigned*>(lds + kPairTableBytes + WAVES * kTileN * 4);
  if (tid == 0) *ticket = 0u;
  u32x4 w0[NBUF];
  float x0[NBUF / 2];
  load_chunk(u_wave, w0, x0);
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (!(ABL & 4)) {
    char* dst = lds + (8 * st_half + 16 * st_ib) * 256 + st_jp * 128 + st_s * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x2*>(dst + i * 256) = f32x2{ea[i], eb};
  }
  f32x2 acc[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
  const bool odd = (grp & 1) != 0;
  const uint32_t off_first = 8u * (uint32_t)i16 + (odd ? 128u : 0u), off_second = 8u * (uint32_t)i16 + (odd ? 0u : 128u);
  __syncthreads();  // tables visible

  auto decode_chunk = [&](int u, const u32x4 (&w)[NBUF], const float (&xs)[NBUF / 2]) __attribute__((always_inline)) {
#pragma unroll
    for (int s2 = 0; s2 < NBUF / 2; ++s2) {
      const int ua = u + 2 * s2 * STEP, ub = ua + STEP;
      const float xa = (ua + grp < u_end) ? xs[s2] : 0.f, xbv = (ub + grp < u_end) ? xs[s2] : 0.f;  // ragged slice end: zero x
      if (ua < u_end) step4_pair<0, ABL>(w[2 * s2], xa, odd, off_first, off_second, acc);
      if (ub < u_end) step4_pair<8, ABL>(w[2 * s2 + 1], xbv, odd, off_first, off_second, acc);
    }
  };
  decode_chunk(u_wave, w0, x0);
  for (int u0 = u_wave + NBUF * STEP; u0 < u_end; u0 += NBUF * STEP) {  // scalar loop
    u32x4 w[NBUF];
    float xs[NBUF / 2];
    load_chunk(u0, w, xs);
    __builtin_amdgcn_sched_barrier(0);
    decode_chunk(u0, w, xs);
  }
  if constexpr (ABL & 8) {
    if (acc[0].x + acc[0].y + acc[1].x + acc[1].y == 12345.678f) y[0] = 1.f;
    return;
  }

  // ---- fold the 4 lane rows, park in this wave's slab, take a ticket; the last wave sums the slabs ----
  if (odd) {  // (odd rows accumulated their pairs in the opposite order)
    const f32x2 tmp = acc[0];
    acc[0] = acc[1];
    acc[1] = tmp;
  }
  float col[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float a = (j & 1) ? acc[j >> 1].y : acc[j >> 1].x;
    a += __shfl_xor(a, 16, 64);
    a += __shfl_xor(a, 32, 64);
    col[j] = a;
  }
  if (grp == 0) *reinterpret_cast<f32x4*>(slabs + wave * kTileN + 4 * i16) = f32x4{col[0], col[1], col[2], col[3]};
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  unsigned tk = 0;
  if (lane == 0) tk = atomicAdd(ticket, 1u);
  tk = __builtin_amdgcn_readfirstlane(tk);
  if (tk != WAVES - 1) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const int c = col0 + lane;
  if (c < N) {
    float sum = 0.f;
#pragma unroll
    for (int wv = 0; wv < WAVES; ++wv) sum += slabs[wv * kTileN + lane];
    acc_add(y + c, sum);
  }
}

template <int ABL>
__global__ void __launch_bounds__(kPairWaves * 64, 8)
sqllm_pair4_matvec(const float* x, const GroupArgs ga) {
  __shared__ __attribute__((aligned(16))) char lds[kPairLdsBytes > 4 * cmax(2 * kCsrSpanMax, kTopxLds) ? kPairLdsBytes : 4 * cmax(2 * kCsrSpanMax, kTopxLds)];
  // one round of scalar loads for the block table and segment 0, a second one for a later segment (see sqllm_fused_matvec)
  Segment sg = ga.seg[0];
  const int n_seg = ga.n_seg, blk1 = ga.block0[1], blk2 = ga.block0[2], blk3 = ga.block0[3];
  asm volatile("" ::SQLLM_SEG_OPERANDS(sg), "s"(x), "s"(n_seg), "s"(blk1), "s"(blk2), "s"(blk3));
  __builtin_amdgcn_sched_barrier(0);
  int s = 0, base = 0;
  if (n_seg > 1 && (int)blockIdx.x >= blk1) { s = 1; base = blk1; }
  if (n_seg > 2 && (int)blockIdx.x >= blk2) { s = 2; base = blk2; }
  if (n_seg > 3 && (int)blockIdx.x >= blk3) { s = 3; base = blk3; }
  s = __builtin_amdgcn_readfirstlane(s);
  if (s != 0) {
    sg = ga.seg[s];
    asm volatile("" ::SQLLM_SEG_OPERANDS(sg));
  }
  const KernelGeom& gm = sg.gm;
  const int bid = blockIdx.x - base;
  const int d = bid - gm.dense_block0;
  if (d >= 0 && d < gm.dense_blocks) {
    dense_role_pair4<ABL>(x, reinterpret_cast<const char*>(sg.q), sg.y, sg.lut, gm.K, gm.N, d, gm.col_tiles, gm.units_total,
                          gm.units_per_wg, lds);
    return;
  }
  // sparse roles: the fused kernels' 8-wave code in the first half of the workgroup (waves that have ended do
  // not take part in the barriers of the others)
  if (threadIdx.x >= 512) return;
  constexpr int T = 512;
  float* fl = reinterpret_cast<float*>(lds);
  if (bid >= 0 && bid < gm.csr_blocks) {
    csr_role<T, 1, float, float>(x, sg.y, sg.rows, sg.cols, sg.vals, gm.nnz, gm.K, gm.N, 0, 1, bid, fl, nullptr, gm.sparse_last >> 1);
  } else if (bid >= gm.csr_blocks && bid < gm.csr_blocks + gm.topx_blocks) {
    topx_role<T, float, float>(x, sg.y, sg.full_rows, sg.full_idx, gm.topX, gm.K, gm.N, 0, 1, bid - gm.csr_blocks, fl);
  }
}

template <int ABL>
static hipError_t launch_pair_inst(const LaunchArgs& a, hipStream_t stream) {
  dim3 grid(a.ga.block0[a.ga.n_seg]);
  auto kern = sqllm_pair4_matvec<ABL>;
  const float* x = static_cast<const float*>(a.x);
  if (a.ev_start || a.ev_stop) hipExtLaunchKernelGGL(kern, grid, dim3(kPairWaves * 64), 0, stream, a.ev_start, a.ev_stop, 0, x, a.ga);
  else hipLaunchKernelGGL(kern, grid, dim3(kPairWaves * 64), 0, stream, x, a.ga);
  return hipGetLastError();
}

hipError_t launch_pair4(const LaunchArgs& a, hipStream_t stream) {
#ifdef SQLLM_ABLATION_BUILD
  switch (a.ablate) {
    case 2: return launch_pair_inst<2>(a, stream);
    case 4: return launch_pair_inst<4>(a, stream);
    case 8: return launch_pair_inst<8>(a, stream);
    case 14: return launch_pair_inst<14>(a, stream);
    default: break;
  }
#endif
  return launch_pair_inst<0>(a, stream);
}

}  // namespace sqllm
