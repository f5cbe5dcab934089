// sqllm_decode.h -- device-side building blocks shared by the kernels of sqllm_kernels.hip and
// sqllm_stream.hip: vector types, the 3-/4-bit field extraction and lookup/FMA stages of the dense
// decode, the fused linear's fixed-point completion words, global-address-space atomics.
// (Everything here is __device__ __forceinline__; the arithmetic contract is the reference's,
// squeezellm/quant_cuda_kernel.cu:741-880.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "sqllm_kernels.h"

// measurement switches (guarded in sqllm_kernels.h: measurement builds only); production values:
#ifndef SQLLM_PAIR3
#define SQLLM_PAIR3 1  // 0 (measurement builds): 3-bit batch-1 decode with one lookup per weight
#endif
#ifndef SQLLM_PAIR3_NOCONFLICT
// 1 (measurement builds, WRONG RESULTS): the 3-bit pair lookups take their entry's parity from the
// lane row instead of from the data, so the two lane rows of a half-wave can never meet on a bank --
// same instruction count, zero bank conflicts: the A/B that prices the conflicts of the real layout
#define SQLLM_PAIR3_NOCONFLICT 0
#endif
#ifndef SQLLM_MFMA_VAR
#define SQLLM_MFMA_VAR 0
#endif
#ifndef SQLLM_MFMA_FAKE
#define SQLLM_MFMA_FAKE 0  // 1 (measurement builds, wrong results): the wide-batch kernel without its matrix instructions
#endif
#ifndef SQLLM_HALF_STAGES
#define SQLLM_HALF_STAGES 1  // 0 (measurement builds): whole-stage decode, 32 live lookups
#endif


namespace sqllm {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// A "unit" is the smallest piece of K that can be decoded on its own: one qweight row (8 k's) for
// 4-bit, three rows (32 k's, squeezellm/quant.py:185-203) for 3-bit.
template <int BITS> struct Fmt;
template <> struct Fmt<4> {
  static constexpr int kLut = 16;   // codebook entries per channel
  static constexpr int kRows = 1;   // qweight rows per unit
  static constexpr int kK = 8;      // k's per unit
};
template <> struct Fmt<3> {
  static constexpr int kLut = 8;
  static constexpr int kRows = 3;
  static constexpr int kK = 32;
};
// element type of vec: fp32 behind the reference operator names, fp16 for the fused linear
template <bool LIN> struct XType { using type = float; };
template <> struct XType<true> { using type = _Float16; };
// accumulator word: the caller's fp32 `mul`, or the fused linear's fixed-point workspace plane
template <bool LIN> struct AccType { using type = float; };
template <> struct AccType<true> { using type = unsigned long long; };

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// broadcast lane P of each 16-lane DPP row to the whole row
template <int P>
__device__ __forceinline__ float row_bcast(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + P, 0xf, 0xf, true));
}

// generic DPP move: lanes the control leaves without a source (or the row mask disables) get `old`.
// Controls used: 0x110 + n = row_shr:n, 0x130 / 0x138 = wave_shl:1 / wave_shr:1, 0x142 / 0x143 = row_bcast:15 / 31
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int v, int old) {
  return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}

// ------------------------------------------------------------------------------------------------
// Field extraction.  Both return the index already multiplied by 128 (bits [7, 7+BITS)), ready to be
// OR-ed into an LDS byte address (one entry row of a sub-table is 32 slots x 4 B = 128 B).
//
// 3-bit: the three rows of a unit form one little-endian 96-bit stream in which weight k occupies
// bits [3k, 3k+3): row0 bits 0-29 are k0..9, row0[30:31] + row1[0] are k10, row1[1:30] are k11..20,
// row1[31] + row2[0:1] are k21, row2[2:31] are k22..31 -- exactly the layout pack2 writes
// (squeezellm/quant.py:185-203) and the reference decodes with its two "straddler" expressions
// (quant_cuda_kernel.cu:792, :809).
// ------------------------------------------------------------------------------------------------
template <int KIDX>
__device__ __forceinline__ uint32_t field3_x128(uint32_t t0, uint32_t t1, uint32_t t2) {
  constexpr int bit = 3 * KIDX;
  constexpr int w = bit >> 5;
  constexpr int o = bit & 31;
  const uint32_t lo = (w == 0) ? t0 : (w == 1) ? t1 : t2;
  uint32_t f;
  if constexpr (o <= 29) {
    if constexpr (o > 7) f = lo >> (o - 7);
    else if constexpr (o < 7) f = lo << (7 - o);
    else f = lo;
  } else {
    const uint32_t hi = (w == 0) ? t1 : t2;
    f = __builtin_amdgcn_alignbit(hi, lo, o) << 7;
  }
  return f & 0x380u;
}

template <int P>
__device__ __forceinline__ uint32_t field4_x128(uint32_t t) {
  uint32_t f;
  if constexpr (4 * P > 7) f = t >> (4 * P - 7);
  else f = t << (7 - 4 * P);
  return f & 0x780u;
}

__device__ __forceinline__ float lds_read_f32(uint32_t byte_addr) {
  return *reinterpret_cast<const float __attribute__((address_space(3)))*>(byte_addr);
}

// ABL (ablation bits, measurement builds only; 0 in production):
//   1 = no LDS lookup (value = address bits), 2 = pure stream (no decode, no FMA),
//   4 = no codebook staging, 8 = no epilogue (reduction + atomics)
template <int ABL>
__device__ __forceinline__ float lookup(uint32_t a) {
  if constexpr (ABL & 1) return __builtin_bit_cast(float, a);
  else return lds_read_f32(a);
}

// Pin values: nothing that consumes them can be placed above this point, and the statement is
// ordered against the other pins / scheduling fences.  Keeps each decode stage's shifts from being
// hoisted to the top of the loop body by instruction selection (which then spills).
#define SQLLM_PIN4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))

// ------------------------------------------------------------------------------------------------
// One decode stage = 8 consecutive k's of this lane's 4 columns: 32 lookups, then 32 FMAs against
// the 8 broadcast x values; the scheduling fence closes the stage.
//   4-bit: a stage is one qweight row (the lane's uint4).
//   3-bit: a unit has 4 stages Q = 0..3 (k = 8Q .. 8Q+7) over the three uint4 of the unit.
// XL = lane (within the 16-lane row) holding x of the stage's first k.
// ------------------------------------------------------------------------------------------------
template <int BT, int XL, int ABL>
__device__ __forceinline__ void fma_stage(const float (&v)[4][8], const float (&xv)[BT], f32x2 (&acc)[2][BT]) {
  // packed fp32 FMAs (v_pk_fma_f32: two columns per instruction, x splat through op_sel): the
  // kernel is issue-bound and this halves its FMA instructions
#pragma unroll
  for (int b = 0; b < BT; ++b) {
    const float x0 = row_bcast<XL + 0>(xv[b]), x1 = row_bcast<XL + 1>(xv[b]);
    const float x2 = row_bcast<XL + 2>(xv[b]), x3 = row_bcast<XL + 3>(xv[b]);
    const float x4 = row_bcast<XL + 4>(xv[b]), x5 = row_bcast<XL + 5>(xv[b]);
    const float x6 = row_bcast<XL + 6>(xv[b]), x7 = row_bcast<XL + 7>(xv[b]);
#define SQLLM_PKFMA(I, X) a = __builtin_elementwise_fma(f32x2{v[2 * jp][I], v[2 * jp + 1][I]}, f32x2{X, X}, a)
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      f32x2 a = acc[jp][b];
      SQLLM_PKFMA(0, x0); SQLLM_PKFMA(1, x1); SQLLM_PKFMA(2, x2); SQLLM_PKFMA(3, x3);
      SQLLM_PKFMA(4, x4); SQLLM_PKFMA(5, x5); SQLLM_PKFMA(6, x6); SQLLM_PKFMA(7, x7);
      acc[jp][b] = a;
    }
#undef SQLLM_PKFMA
  }
}

// 4-bit step: one qweight row of this lane's 4 columns x 8 weights.
// Address generation is the VALU hot spot (the kernel is VALU-bound: every wave64 VALU op costs 4
// cycles of its SIMD), so it is done with ONE v_perm_b32 per weight: the word is first split into
// nibble-bytes  lo = w & 0x0F0F0F0F (nibbles 0,2,4,6)  and  hi = (w >> 4) & 0x0F0F0F0F (1,3,5,7)
// -- 3 ops per 8 weights -- and the 4-bit table uses a 256-byte entry stride, so the LDS byte
// address of a lookup is simply  [byte1 = nibble, byte0 = 4 * lane] : a byte permute of (nibble
// word, lane-offset word).  The sub-table of column j sits at a constant +4096 j, which folds into
// the ds_read's immediate offset.  XL = lane of the 16-lane row holding x of the row's first k.
template <int BT, int XL, int ABL>
__device__ __forceinline__ void step4(const u32x4& slot, const float (&xslot)[BT], bool valid,
                                      uint32_t lane_off, f32x2 (&acc)[2][BT]) {
  uint32_t t[4] = {slot.x, slot.y, slot.z, slot.w};
  float xv[BT];
  SQLLM_PIN4(t[0], t[1], t[2], t[3]);
#pragma unroll
  for (int b = 0; b < BT; ++b) xv[b] = valid ? xslot[b] : 0.f;
  if constexpr (ABL & 2) {
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      acc[0][b].x += __builtin_bit_cast(float, t[0] ^ t[1]) * xv[b];
      acc[1][b].x += __builtin_bit_cast(float, t[2] ^ t[3]) * xv[b];
    }
    return;
  }
  float v[4][8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t lo = t[j] & 0x0F0F0F0Fu;
    const uint32_t hi = (t[j] >> 4) & 0x0F0F0F0Fu;
    // columns j and j^1 share 256-byte entry rows (even column in the low 128 bytes, odd in the
    // high), the pair (j >> 1) selects the 4 KiB half: both fold into the ds_read immediate.
    // selector bytes (LSB first): byte0 <- lane_off.byte0, byte1 <- nibble word byte k, bytes 2,3 <- 0
    constexpr int kNoOff = 0;
    const int off = (j >> 1) * 4096 + (j & 1) * 128 + kNoOff;
    v[j][0] = lookup<ABL>(__builtin_amdgcn_perm(lo, lane_off, 0x0C0C0400u) + off);
    v[j][1] = lookup<ABL>(__builtin_amdgcn_perm(hi, lane_off, 0x0C0C0400u) + off);
    v[j][2] = lookup<ABL>(__builtin_amdgcn_perm(lo, lane_off, 0x0C0C0500u) + off);
    v[j][3] = lookup<ABL>(__builtin_amdgcn_perm(hi, lane_off, 0x0C0C0500u) + off);
    v[j][4] = lookup<ABL>(__builtin_amdgcn_perm(lo, lane_off, 0x0C0C0600u) + off);
    v[j][5] = lookup<ABL>(__builtin_amdgcn_perm(hi, lane_off, 0x0C0C0600u) + off);
    v[j][6] = lookup<ABL>(__builtin_amdgcn_perm(lo, lane_off, 0x0C0C0700u) + off);
    v[j][7] = lookup<ABL>(__builtin_amdgcn_perm(hi, lane_off, 0x0C0C0700u) + off);
  }
  fma_stage<BT, XL, ABL>(v, xv, acc);
  __builtin_amdgcn_sched_barrier(0);
}

// packed FMAs of ONE column pair: vp[i] = the values of weight k = i of the pair's two columns
template <int BT, int XL>
__device__ __forceinline__ void fma_pair(const f32x2 (&vp)[8], const float (&xv)[BT], f32x2 (&acc)[BT]) {
#pragma unroll
  for (int b = 0; b < BT; ++b) {
    const float x0 = row_bcast<XL + 0>(xv[b]), x1 = row_bcast<XL + 1>(xv[b]), x2 = row_bcast<XL + 2>(xv[b]), x3 = row_bcast<XL + 3>(xv[b]);
    const float x4 = row_bcast<XL + 4>(xv[b]), x5 = row_bcast<XL + 5>(xv[b]), x6 = row_bcast<XL + 6>(xv[b]), x7 = row_bcast<XL + 7>(xv[b]);
    f32x2 a = acc[b];
    a = __builtin_elementwise_fma(vp[0], f32x2{x0, x0}, a);
    a = __builtin_elementwise_fma(vp[1], f32x2{x1, x1}, a);
    a = __builtin_elementwise_fma(vp[2], f32x2{x2, x2}, a);
    a = __builtin_elementwise_fma(vp[3], f32x2{x3, x3}, a);
    a = __builtin_elementwise_fma(vp[4], f32x2{x4, x4}, a);
    a = __builtin_elementwise_fma(vp[5], f32x2{x5, x5}, a);
    a = __builtin_elementwise_fma(vp[6], f32x2{x6, x6}, a);
    a = __builtin_elementwise_fma(vp[7], f32x2{x7, x7}, a);
    acc[b] = a;
  }
}

// Half-stage variant of the 4-bit step: one column PAIR at a time -- 16 lookups, then
// their 8 packed FMAs -- so that only 16 lookup registers are live and the kernel fits 64 VGPRs
// (four 8-wave workgroups per CU).
template <int BT, int XL, int ABL>
__device__ __forceinline__ void step4_half(const u32x4& slot, const float (&xslot)[BT], bool valid,
                                           uint32_t lane_off, f32x2 (&acc)[2][BT]) {
  uint32_t t[4] = {slot.x, slot.y, slot.z, slot.w};
  SQLLM_PIN4(t[0], t[1], t[2], t[3]);
  float xv[BT];
#pragma unroll
  for (int b = 0; b < BT; ++b) xv[b] = valid ? xslot[b] : 0.f;
  if constexpr (ABL & 2) {  // measurement: pure stream, no decode (the whole-stage step4 had this; the half-stage one did not)
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      acc[0][b].x += __builtin_bit_cast(float, t[0] ^ t[1]) * xv[b];
      acc[1][b].x += __builtin_bit_cast(float, t[2] ^ t[3]) * xv[b];
    }
    return;
  }
#pragma unroll
  for (int jp = 0; jp < 2; ++jp) {
    f32x2 vp[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int j = 2 * jp + h;
      const uint32_t lo = t[j] & 0x0F0F0F0Fu;
      const uint32_t hi = (t[j] >> 4) & 0x0F0F0F0Fu;
      const int off = jp * 4096 + h * 128;
      float e[8];
      e[0] = lookup<ABL>(__builtin_amdgcn_perm(lo, lane_off, 0x0C0C0400u) + off);
      e[1] = lookup<ABL>(__builtin_amdgcn_perm(hi, lane_off, 0x0C0C0400u) + off);
      e[2] = lookup<ABL>(__builtin_amdgcn_perm(lo, lane_off, 0x0C0C0500u) + off);
      e[3] = lookup<ABL>(__builtin_amdgcn_perm(hi, lane_off, 0x0C0C0500u) + off);
      e[4] = lookup<ABL>(__builtin_amdgcn_perm(lo, lane_off, 0x0C0C0600u) + off);
      e[5] = lookup<ABL>(__builtin_amdgcn_perm(hi, lane_off, 0x0C0C0600u) + off);
      e[6] = lookup<ABL>(__builtin_amdgcn_perm(lo, lane_off, 0x0C0C0700u) + off);
      e[7] = lookup<ABL>(__builtin_amdgcn_perm(hi, lane_off, 0x0C0C0700u) + off);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (h) vp[i].y = e[i]; else vp[i].x = e[i];
      }
    }
    fma_pair<BT, XL>(vp, xv, acc[jp]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int BT, int Q, int ABL, bool HALF = false>
__device__ __forceinline__ void stage3(const uint32_t (&t0)[4], const uint32_t (&t1)[4], const uint32_t (&t2)[4],
                                       const uint32_t (&tb)[4], const float (&xlo)[BT], const float (&xhi)[BT],
                                       f32x2 (&acc)[2][BT]) {
  if constexpr (HALF) {  // one column pair at a time: 16 live lookups (see step4_half)
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      f32x2 vp[8];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int j = 2 * jp + h;
        float e[8];
        e[0] = lookup<ABL>(tb[j] | field3_x128<8 * Q + 0>(t0[j], t1[j], t2[j]));
        e[1] = lookup<ABL>(tb[j] | field3_x128<8 * Q + 1>(t0[j], t1[j], t2[j]));
        e[2] = lookup<ABL>(tb[j] | field3_x128<8 * Q + 2>(t0[j], t1[j], t2[j]));
        e[3] = lookup<ABL>(tb[j] | field3_x128<8 * Q + 3>(t0[j], t1[j], t2[j]));
        e[4] = lookup<ABL>(tb[j] | field3_x128<8 * Q + 4>(t0[j], t1[j], t2[j]));
        e[5] = lookup<ABL>(tb[j] | field3_x128<8 * Q + 5>(t0[j], t1[j], t2[j]));
        e[6] = lookup<ABL>(tb[j] | field3_x128<8 * Q + 6>(t0[j], t1[j], t2[j]));
        e[7] = lookup<ABL>(tb[j] | field3_x128<8 * Q + 7>(t0[j], t1[j], t2[j]));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (h) vp[i].y = e[i]; else vp[i].x = e[i];
        }
      }
      if constexpr (Q < 2) fma_pair<BT, 8 * Q>(vp, xlo, acc[jp]);
      else fma_pair<BT, 8 * (Q - 2)>(vp, xhi, acc[jp]);
      __builtin_amdgcn_sched_barrier(0);
    }
    return;
  }
  float v[4][8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j][0] = lookup<ABL>(tb[j] | field3_x128<8 * Q + 0>(t0[j], t1[j], t2[j]));
    v[j][1] = lookup<ABL>(tb[j] | field3_x128<8 * Q + 1>(t0[j], t1[j], t2[j]));
    v[j][2] = lookup<ABL>(tb[j] | field3_x128<8 * Q + 2>(t0[j], t1[j], t2[j]));
    v[j][3] = lookup<ABL>(tb[j] | field3_x128<8 * Q + 3>(t0[j], t1[j], t2[j]));
    v[j][4] = lookup<ABL>(tb[j] | field3_x128<8 * Q + 4>(t0[j], t1[j], t2[j]));
    v[j][5] = lookup<ABL>(tb[j] | field3_x128<8 * Q + 5>(t0[j], t1[j], t2[j]));
    v[j][6] = lookup<ABL>(tb[j] | field3_x128<8 * Q + 6>(t0[j], t1[j], t2[j]));
    v[j][7] = lookup<ABL>(tb[j] | field3_x128<8 * Q + 7>(t0[j], t1[j], t2[j]));
  }
  if constexpr (Q < 2) fma_stage<BT, 8 * Q, ABL>(v, xlo, acc);
  else fma_stage<BT, 8 * (Q - 2), ABL>(v, xhi, acc);
  __builtin_amdgcn_sched_barrier(0);
}

template <int BT, int ABL, bool HALF = false>
__device__ __forceinline__ void step3(const u32x4 (&slot)[3], const float (&xslot0)[BT], const float (&xslot1)[BT],
                                      bool valid, const uint32_t (&tb)[4], f32x2 (&acc)[2][BT]) {
  uint32_t t0[4] = {slot[0].x, slot[0].y, slot[0].z, slot[0].w};
  uint32_t t1[4] = {slot[1].x, slot[1].y, slot[1].z, slot[1].w};
  uint32_t t2[4] = {slot[2].x, slot[2].y, slot[2].z, slot[2].w};
  float xlo[BT], xhi[BT];
  SQLLM_PIN4(t0[0], t0[1], t0[2], t0[3]);
  SQLLM_PIN4(t1[0], t1[1], t1[2], t1[3]);
  SQLLM_PIN4(t2[0], t2[1], t2[2], t2[3]);
#pragma unroll
  for (int b = 0; b < BT; ++b) {
    xlo[b] = valid ? xslot0[b] : 0.f;
    xhi[b] = valid ? xslot1[b] : 0.f;
  }
  stage3<BT, 0, ABL, HALF>(t0, t1, t2, tb, xlo, xhi, acc);
  stage3<BT, 1, ABL, HALF>(t0, t1, t2, tb, xlo, xhi, acc);
  stage3<BT, 2, ABL, HALF>(t0, t1, t2, tb, xlo, xhi, acc);
  stage3<BT, 3, ABL, HALF>(t0, t1, t2, tb, xlo, xhi, acc);
}

// ------------------------------------------------------------------------------------------------
// 3-bit PAIR decode (SQLLM_PAIR3; batch tile 1): the kernel is bound by the SUM of its vector and
// LDS instructions, and the plain 3-bit path spends 4.1 of them per weight.
// Here a column's codebook is staged as a 64-entry table of PAIRS -- entry i0 + 8 * i1 holds
// (lut[i0], lut[i1]) -- so that one ds_read_b64, addressed by the 6-bit field of two consecutive
// weights (k, k+1), returns both values, and one packed FMA multiplies them by (x[k], x[k+1]):
// 2 address ops + 1 lookup + 1 FMA + 1/2 broadcast per TWO weights.  The accumulator of a column
// is a float2 (even k, odd k), summed at the end.  32 KB of tables per 64-column tile.
// ------------------------------------------------------------------------------------------------
template <int M>  // pair M of a unit: weights k = 2M, 2M+1 = bits [6M, 6M+6) of the 96-bit stream; result << 7
__device__ __forceinline__ uint32_t field6_x128(uint32_t t0, uint32_t t1, uint32_t t2) {
  constexpr int bit = 6 * M;
  constexpr int w = bit >> 5;
  constexpr int o = bit & 31;
  const uint32_t lo = (w == 0) ? t0 : (w == 1) ? t1 : t2;
  uint32_t f;
  if constexpr (o <= 26) {
    if constexpr (o > 7) f = lo >> (o - 7);
    else if constexpr (o < 7) f = lo << (7 - o);
    else f = lo;
  } else {  // pairs 5 and 10 straddle a dword boundary
    const uint32_t hi = (w == 0) ? t1 : t2;
    f = __builtin_amdgcn_alignbit(hi, lo, o) << 7;
  }
#if SQLLM_PAIR3_NOCONFLICT
  return f & 0x1F00u;  // measurement build: entry parity comes from the lane row (see tb[] in dense_role)
#else
  return f & 0x1F80u;
#endif
}

__device__ __forceinline__ f32x2 lds_read_f32x2(uint32_t byte_addr) {
  return *reinterpret_cast<const f32x2 __attribute__((address_space(3)))*>(byte_addr);
}

// 8 pairs (16 k's) of ONE column: 8 lookups live at a time
template <int H>
__device__ __forceinline__ void stage3_pair(uint32_t t0, uint32_t t1, uint32_t t2, uint32_t tbj,
                                            const float (&xb)[16], f32x2& acc) {
  f32x2 v[8];
  v[0] = lds_read_f32x2(tbj | field6_x128<8 * H + 0>(t0, t1, t2));
  v[1] = lds_read_f32x2(tbj | field6_x128<8 * H + 1>(t0, t1, t2));
  v[2] = lds_read_f32x2(tbj | field6_x128<8 * H + 2>(t0, t1, t2));
  v[3] = lds_read_f32x2(tbj | field6_x128<8 * H + 3>(t0, t1, t2));
  v[4] = lds_read_f32x2(tbj | field6_x128<8 * H + 4>(t0, t1, t2));
  v[5] = lds_read_f32x2(tbj | field6_x128<8 * H + 5>(t0, t1, t2));
  v[6] = lds_read_f32x2(tbj | field6_x128<8 * H + 6>(t0, t1, t2));
  v[7] = lds_read_f32x2(tbj | field6_x128<8 * H + 7>(t0, t1, t2));
  f32x2 a = acc;
  a = __builtin_elementwise_fma(v[0], f32x2{xb[0], xb[1]}, a);
  a = __builtin_elementwise_fma(v[1], f32x2{xb[2], xb[3]}, a);
  a = __builtin_elementwise_fma(v[2], f32x2{xb[4], xb[5]}, a);
  a = __builtin_elementwise_fma(v[3], f32x2{xb[6], xb[7]}, a);
  a = __builtin_elementwise_fma(v[4], f32x2{xb[8], xb[9]}, a);
  a = __builtin_elementwise_fma(v[5], f32x2{xb[10], xb[11]}, a);
  a = __builtin_elementwise_fma(v[6], f32x2{xb[12], xb[13]}, a);
  a = __builtin_elementwise_fma(v[7], f32x2{xb[14], xb[15]}, a);
  acc = a;
  __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void step3_pair(const u32x4 (&slot)[3], float xslot0, float xslot1, bool valid,
                                           const uint32_t (&tb)[4], f32x2 (&accp)[4]) {
  uint32_t t0[4] = {slot[0].x, slot[0].y, slot[0].z, slot[0].w};
  uint32_t t1[4] = {slot[1].x, slot[1].y, slot[1].z, slot[1].w};
  uint32_t t2[4] = {slot[2].x, slot[2].y, slot[2].z, slot[2].w};
  SQLLM_PIN4(t0[0], t0[1], t0[2], t0[3]);
  SQLLM_PIN4(t1[0], t1[1], t1[2], t1[3]);
  SQLLM_PIN4(t2[0], t2[1], t2[2], t2[3]);
  const float xlo = valid ? xslot0 : 0.f, xhi = valid ? xslot1 : 0.f;
#define SQLLM_XB16(X) {row_bcast<0>(X), row_bcast<1>(X), row_bcast<2>(X), row_bcast<3>(X), row_bcast<4>(X), row_bcast<5>(X), \
                       row_bcast<6>(X), row_bcast<7>(X), row_bcast<8>(X), row_bcast<9>(X), row_bcast<10>(X), row_bcast<11>(X), \
                       row_bcast<12>(X), row_bcast<13>(X), row_bcast<14>(X), row_bcast<15>(X)}
  {
    const float xb[16] = SQLLM_XB16(xlo);  // x of k = 0..15 of the unit, broadcast along the 16-lane row
#pragma unroll
    for (int j = 0; j < 4; ++j) stage3_pair<0>(t0[j], t1[j], t2[j], tb[j], xb, accp[j]);
  }
  {
    const float xb[16] = SQLLM_XB16(xhi);  // k = 16..31
#pragma unroll
    for (int j = 0; j < 4; ++j) stage3_pair<1>(t0[j], t1[j], t2[j], tb[j], xb, accp[j]);
  }
#undef SQLLM_XB16
}

// ------------------------------------------------------------------------------------------------
// Fused-linear completion (sqllm_linear_f16: fp16 in, fp16 out, bias, no launches around the op --
// the reference wraps every op in a zeros/clone, an x.float() and a y.to(fp16) kernel,
// squeezellm/quant.py:214-223,311-312).
//
// The roles accumulate into a plane of 64-bit words in the caller's workspace, all zero between
// launches.  A word is  count * 2^55 + S  with S the column's sum in signed fixed point (2^-28
// units): integer adds commute, so ONE returning atomic add both deposits a contribution and
// tells the contributor how many have arrived.  How many a column will receive is known to every
// contributor without communication:
//     every dense K slice of the column's tile            -> k_slices
//   + every CSR chunk that holds part of the column's row -> from rows[c], rows[c+1] alone
// (the top-X rows are folded into the dense workgroups of the tiles that own their columns, see
// dense_role).  Whoever deposits the last contribution owns the column: bias, fp16 store, word
// back to zero.  The critical path of a workgroup grows by one atomic round trip; there are no
// fences (an agent-scope release/acquire pair costs an L2 write-back and an L2 invalidate per
// workgroup here: measured +4.5 us per launch) and no launch-wide counter (a last-arriver that
// must then touch all N columns measured +4-11 us per launch).
//
// Contributions are clamped to +-2^17 (twice the largest finite fp16) so that the at most 63 of
// them a column can receive stay inside the 55-bit field; sums beyond that are not finite in
// fp16 anyway.  Rounding: 2^-28 absolute per contribution, far below one fp16 ulp of any normal
// fp16 result.
//
// Non-finite values (round 3).  The reference's fp32 atomics carry NaN / Inf through to the output
// (squeezellm/quant.py:214-223: zeros, op, cast); an integer sum cannot, so the word's two top bits
// are STICKY FLAGS, set with an atomic OR by a contributor whose partial sum is not finite BEFORE
// its counted add (atomics on one address are totally ordered, so whoever completes the count sees
// every flag): bit 63 = a NaN was contributed, bit 62 = a +inf, bit 61 = a -inf (infinities of both
// signs make a NaN, as inf - inf does); a non-finite value adds nothing to the sum.  The count keeps
// bits 55-60 (at most 63 contributions).
// (Only the wide-span CSR fallback adds values uncounted; while such a word is transiently negative an
// OR may be lost -- a NaN is then reported as a finite number, as before round 3.)
// ------------------------------------------------------------------------------------------------
typedef unsigned long long u64;
constexpr int kFixShift = 28;
constexpr int kCountShift = 55;
constexpr u64 kCountUnit = 1ull << kCountShift;
constexpr u64 kNanFlag = 1ull << 63, kPosInfFlag = 1ull << 62, kNegInfFlag = 1ull << 61, kFlagMask = kNanFlag | kPosInfFlag | kNegInfFlag;

__device__ __forceinline__ u64 to_fixed(float v) {
  v = (__builtin_fabsf(v) <= 3.402823466e38f) ? __builtin_fminf(__builtin_fmaxf(v, -131072.f), 131072.f) : 0.f;  // (NaN / inf ride in the flags)
  return (u64)(long long)__builtin_rintf(v * (float)(1 << kFixShift));
}

// set the word's sticky flag if `v` is not finite (call before the add that deposits v)
__device__ __forceinline__ void flag_nonfinite(u64* word, float v) {
  if (!(__builtin_fabsf(v) <= 3.402823466e38f))
    __hip_atomic_fetch_or(reinterpret_cast<__attribute__((address_space(1))) u64*>(reinterpret_cast<uintptr_t>(word)),
                          (v != v) ? kNanFlag : (v > 0.f ? kPosInfFlag : kNegInfFlag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// CSR chunks (kCsrChunk consecutive non-zeros each) holding part of a row that spans [r0, r1)
__device__ __forceinline__ int csr_chunks_of_row(int r0, int r1) {
  return r1 > r0 ? (r1 - 1) / kCsrChunk - r0 / kCsrChunk + 1 : 0;
}

// `total` = the word after this thread's own counted add.  Finishes the column if that add was the
// last of the `target` contributions.
__device__ __forceinline__ void column_done(const Segment& sg, u64* word, u64 total, unsigned target,
                                            size_t at, int c) {
  const u64 flags = total & kFlagMask;
  total &= ~kFlagMask;
  const u64 count = (total + (kCountUnit >> 1)) >> kCountShift;  // S may be negative: round, do not truncate
  if ((unsigned)count != target) return;
  const long long sfix = (long long)(total - (count << kCountShift));
  float v = (float)sfix * (1.f / (float)(1 << kFixShift));
  if ((flags & kNanFlag) || (flags & (kPosInfFlag | kNegInfFlag)) == (kPosInfFlag | kNegInfFlag)) v = __builtin_nanf("");
  else if (flags & kPosInfFlag) v = __builtin_inff();
  else if (flags & kNegInfFlag) v = -__builtin_inff();
  v += sg.bias ? sg.bias[c] : 0.f;
  reinterpret_cast<_Float16*>(sg.out16)[at] = (_Float16)v;
  atomicExch(word, 0ull);  // result unused: a plain atomic store
}

// accumulate one UNCOUNTED value: fp32 atomic (operator launches) or fixed-point add (fused linear).
// The pointer is cast to the global address space on purpose: through a generic pointer these
// become FLAT atomics, and a flat operation anywhere upstream in the kernel's control-flow graph
// makes the compiler treat vmcnt as out of order -- every later wait for a load turns into
// vmcnt(0), including the codebook staging wait of the dense role (+0.3-0.6 us per launch).
#define SQLLM_GLOBAL(T, p) reinterpret_cast<__attribute__((address_space(1))) T*>(reinterpret_cast<uintptr_t>(p))
__device__ __forceinline__ void acc_add(float* p, float v) {
  __hip_atomic_fetch_add(SQLLM_GLOBAL(float, p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void acc_add(u64* p, float v) {
  flag_nonfinite(p, v);
  __hip_atomic_fetch_add(SQLLM_GLOBAL(u64, p), to_fixed(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// vec element load.  COH (dependency-gated pass, sqllm_pass.hip): the element may have been produced by ANOTHER
// workgroup of the same launch -- with device-scope atomics, performed at the memory side -- so it is read with an
// agent-scope (sc1) load, which is never served from this CU's L1; paired with the producers' atomics and the
// arrival counter this needs no fence on either side (MI355X_MICROARCH.md, "Valid forms": agent atomics both sides).
template <bool COH, typename XT>
__device__ __forceinline__ float ld_x(const XT* p) {
  if constexpr (COH) {
    static_assert(std::is_same<XT, float>::value, "the gated pass runs the fp32 operator ABI");
    return __hip_atomic_load(SQLLM_GLOBAL(const float, p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    return (float)*p;
  }
}

// Force every field of a segment descriptor into registers HERE (an empty asm statement that names the
// value as a scalar INPUT operand: the loads feeding it must have completed; an in/out operand would
// also hide where a pointer came from and turn every access through it into a FLAT instruction): the
// compiler otherwise keeps a pointer per field and loads each one where it is first used, one
// dependent scalar-load round trip (0.15 us) at a time.
// (ONE statement for all of them: every asm statement waits for its own operands, and loads are not
// moved above an earlier volatile asm.)
#define SQLLM_SEG_OPERANDS(sg)                                                                                         \
  "s"(sg.q), "s"(sg.y), "s"(sg.lut), "s"(sg.rows), "s"(sg.cols), "s"(sg.vals), "s"(sg.full_rows), "s"(sg.full_idx),    \
  "s"(sg.bias), "s"(sg.out16), "s"(sg.gm.K), "s"(sg.gm.N), "s"(sg.gm.batch), "s"(sg.gm.col_tiles),                     \
  "s"(sg.gm.units_total), "s"(sg.gm.units_per_wg), "s"(sg.gm.k_slices), "s"(sg.gm.dense_blocks),                       \
  "s"(sg.gm.dense_block0), "s"(sg.gm.csr_blocks), "s"(sg.gm.topx_blocks), "s"(sg.gm.nnz), "s"(sg.gm.topX),             \
  "s"(sg.gm.sparse_last), "s"(sg.gm.dense_prio), "s"(sg.gm.csr_wide)

}  // namespace sqllm
