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
//   [dense_block0, +dense_blocks)       dense tiles: 64 output columns x one K slice
// and every role accumulates into `mul` with fp32 atomics, as the reference does.
//
// Dense tile design (measured choices, see DESIGN.md section "dense kernel"):
//   * qweight is int32 [K/32*bits, N] row-major.  A lane owns 4 adjacent columns and reads them
//     as one 16-byte nontemporal load per qweight row; 16 lanes cover a 64-column tile (a 256-byte
//     row segment -- measured within 5 % of the streaming rate of 1 KiB segments), so one wave
//     load instruction fetches FOUR consecutive rows (32 k's for 4-bit), one per 16-lane row.
//   * a workgroup = one 64-column tile x one K slice, 8 waves.  Narrow tiles make the K slice
//     long (512-2048 k's), which is what amortises the per-workgroup costs: the tile's codebooks
//     are staged once (4 KiB for 4-bit) for 16-64 KiB of weights, and the epilogue issues 64
//     atomics.  (A 256-column tile restaged 16 KiB of codebooks per 16 KiB of weights at the 7B
//     shapes and lost 2 us per launch to it.)
//   * codebooks live in LDS as 4 sub-tables (one per dword of the lane's load) laid out
//     [entry][32 slots], TWO copies of the 16 columns side by side: ds_read_b32 is serviced per
//     half-wave (32 lanes = two 16-lane rows), each row reads its own copy, so a lookup's bank is
//     a function of the lane only and lookups never conflict whatever the indices are.
//   * vec[k]: the 16 lanes of a row all work on the same 8 k's.  Lane i of a row holds
//     x[k0 + (i & 7)] (one coalesced dword load per wave per 32 k's, prefetched with the weights)
//     and the FMAs take it through a DPP row broadcast (v_mov_b32_dpp row_newbcast) -- one extra
//     VALU op per 4 weights, no LDS traffic (the reference reads vec from shared memory once per
//     weight, quant_cuda_kernel.cu:866).
//   * partial sums are folded across the 4 lane rows with two cross-lane adds, across the waves
//     through LDS, and leave the workgroup as one atomic per column.
//   * no MFMA: batch-1 decode is a gather.  What bounds it (measured) is
//     instruction issue: a SIMD starts one wave64 instruction per four cycles, vector OR LDS, so
//     a weight costs the sum of both (4-bit: 2.4 + 1.1).  Two consequences shape the code:
//       - occupancy over ILP: decode stages work on one column pair at a time (16 live lookups),
//         the batch-1 kernels fit 64 VGPRs and run four 8-wave workgroups per CU;
//       - fewer instructions per weight where the format allows it: 3-bit codebooks are staged as
//         64-entry tables of PAIRS and two consecutive weights cost one ds_read_b64 + one packed
//         FMA (2.45 instructions per weight instead of 4.1).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
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
// Contributions are clamped to +-2^17 (twice the largest finite fp16) so that the at most 511 of
// them a column can receive stay inside the 55-bit field; sums beyond that are not finite in
// fp16 anyway.  Rounding: 2^-28 absolute per contribution, far below one fp16 ulp of any normal
// fp16 result.
// ------------------------------------------------------------------------------------------------
typedef unsigned long long u64;
constexpr int kFixShift = 28;
constexpr int kCountShift = 55;
constexpr u64 kCountUnit = 1ull << kCountShift;

__device__ __forceinline__ u64 to_fixed(float v) {
  v = __builtin_fminf(__builtin_fmaxf(v, -131072.f), 131072.f);  // also maps NaN to a bound
  return (u64)(long long)__builtin_rintf(v * (float)(1 << kFixShift));
}

// CSR chunks (kCsrChunk consecutive non-zeros each) holding part of a row that spans [r0, r1)
__device__ __forceinline__ int csr_chunks_of_row(int r0, int r1) {
  return r1 > r0 ? (r1 - 1) / kCsrChunk - r0 / kCsrChunk + 1 : 0;
}

// `total` = the word after this thread's own counted add.  Finishes the column if that add was the
// last of the `target` contributions.
__device__ __forceinline__ void column_done(const Segment& sg, u64* word, u64 total, unsigned target,
                                            size_t at, int c) {
  const u64 count = (total + (kCountUnit >> 1)) >> kCountShift;  // S may be negative: round, do not truncate
  if ((unsigned)count != target) return;
  const long long sfix = (long long)(total - (count << kCountShift));
  const float v = (float)sfix * (1.f / (float)(1 << kFixShift)) + (sg.bias ? sg.bias[c] : 0.f);
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
  __hip_atomic_fetch_add(SQLLM_GLOBAL(u64, p), to_fixed(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
  "s"(sg.gm.sparse_last)

// ------------------------------------------------------------------------------------------------
// Dense epilogue (shared by the dense-role variants): fold the 4 lane rows, then the waves through
// LDS, one atomic per column.  `slabs` = LDS area [WAVES][BT][64] floats followed by the ticket.
// ------------------------------------------------------------------------------------------------
template <int BT, int WAVES, int ABL>
__device__ __forceinline__ void dense_epilogue(const f32x2 (&acc)[2][BT], float* slabs, const float* topx_sum,
                                               bool fold_topx, float* __restrict__ y, int N, int col0, int b0,
                                               int nb, int lane, int wave, const Segment& sg, const Segment* lin
#ifdef SQLLM_ABLATION_BUILD
                                               , unsigned long long* tl
#endif
) {
  const int i16 = lane & 15, grp = lane >> 4;
  if constexpr (ABL & 8) {
    if (acc[0][0].x + acc[0][0].y + acc[1][0].x + acc[1][0].y == 12345.678f) y[0] = 1.f;  // keep the work alive
    return;
  }
  // ---- fold the 4 lane rows, then the waves through LDS (codebooks are dead now); one atomic
  //      per column.  Batch rows go through in chunks of CB so the buffer stays small. ----
  float col[4][BT];  // this lane's four columns, summed over the wave's 4 lane rows
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      float a = (j & 1) ? acc[j >> 1][b].y : acc[j >> 1][b].x;
      a += __shfl_xor(a, 16, 64);
      a += __shfl_xor(a, 32, 64);
      col[j][b] = a;
    }
  if constexpr (ABL & 32) {
    // variant: no cross-wave combine, every wave adds its own 64 partial sums
    if (grp == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = col0 + 4 * i16 + j;
#pragma unroll
        for (int b = 0; b < BT; ++b)
          if (c < N && b < nb) atomicAdd(y + (size_t)(b0 + b) * N + c, col[j][b]);
      }
    }
    return;
  }
  // Barrier-free combine: every wave deposits its 64 x BT partial sums in its own LDS slab (a
  // region the codebooks never occupy, so nobody has to wait for the other waves' lookups), then
  // takes a ticket; the wave that draws the last ticket sums the slabs and issues the atomics.
  // Waves that finish early simply leave.  (LDS operations of a CU execute in issue order and a
  // wave's own LDS operations stay in program order, so the last ticket implies every slab is
  // written; the fence pins the compiler.)  The two-barrier version cost 1-2.5 us per launch.
  float* red = slabs;                                                      // [wave][BT][64]
  unsigned* ticket = reinterpret_cast<unsigned*>(slabs + WAVES * BT * kTileN);
  if (grp == 0) {
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      f32x4 v = {col[0][b], col[1][b], col[2][b], col[3][b]};
      *reinterpret_cast<f32x4*>(red + (wave * BT + b) * kTileN + 4 * i16) = v;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  unsigned t = 0;
  if (lane == 0) t = atomicAdd(ticket, 1u);
  t = __builtin_amdgcn_readfirstlane(t);
  if (t != WAVES - 1) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const int c = col0 + lane;
  if (c < N) {
    u64 total[BT];
    unsigned target = 0;
    if (lin) {  // contributions this column receives: K slices + the CSR chunks its row is spread over
      target = (unsigned)lin->gm.k_slices;
      if (lin->gm.csr_blocks) target += (unsigned)csr_chunks_of_row(lin->rows[c], lin->rows[c + 1]);
    }
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      if (b < nb) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) sum += red[(w * BT + b) * kTileN + lane];
        const size_t at = (size_t)(b0 + b) * N + c;
        if (fold_topx) sum += topx_sum[b * kTileN + lane];
        if (lin) {
          const u64 mine = kCountUnit + to_fixed(sum);
          total[b] = atomicAdd(reinterpret_cast<u64*>(y) + at, mine) + mine;
        } else {
          atomicAdd(y + at, sum);
        }
      }
    }
#ifdef SQLLM_ABLATION_BUILD
    if (tl && lane == 0) tl[3] = __builtin_amdgcn_s_memrealtime();  // atomics issued by the combining wave
#endif
    if (lin) {  // all the round trips are in flight before the first result is looked at
#pragma unroll
      for (int b = 0; b < BT; ++b) {
        const size_t at = (size_t)(b0 + b) * N + c;
        if (b < nb) column_done(*lin, reinterpret_cast<u64*>(y) + at, total[b], target, at, c);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// dense role
//
// Codebook layout in LDS (bytes):  addr(j, idx, slot) = j * SUBB + idx * ESTRIDE + 4 * slot
//   j       = which dword of the lane's 16-byte load (the lane's j-th column)
//   4-bit:  256-byte entry rows (the stride the v_perm address generation wants), each holding the
//           entry of an even column in its low 128 bytes and of the next odd column in its high
//           128 bytes; slot as for 3-bit; two column pairs -> 2 x 4 KiB
//   3-bit:  ESTRIDE = 128, slot = (lane & 15) + 16 * ((lane >> 4) & 1) (two copies: ds_read_b32 is
//           serviced per half-wave of two rows), SUBB = 1024; idx * 128 is OR-ed into the base
//   either way a lookup's bank depends on the lane only: lookups never conflict.
//   (the __shared__ array is the kernel's only LDS object and sits at LDS address 0)
// A step of a wave = one unit per 16-lane row = 4 consecutive units (32 k's for 4-bit, 128 for
// 3-bit); the waves of a workgroup interleave steps, so the workgroup walks its K slice in order.
// A wave issues the loads of a chunk of NBUF steps back to back, then decodes them in arrival
// order (counted vmcnt waits), then loops.  In-flight loads are deliberately NOT carried around
// the loop edge: the kernel is VALU-bound, deeper pipelines measured slower (their copies and
// address arithmetic cost more VALU than the overlap returns), and up to 32 waves per CU at different
// phases keep the memory pipe busy.
// ------------------------------------------------------------------------------------------------
template <int BITS, int BT, int WAVES, int ABL, typename XT, bool HALF = false>
__device__ __forceinline__ void dense_role(const XT* x, const u32x4* q, float* __restrict__ y,
                                           const float* lut, int K, int N, int b0, int nb, int bid,
                                           int n_col_tiles, int units_total, int units_per_wg, float* lds,
                                           const Segment& sg, const Segment* lin) {
  using F = Fmt<BITS>;
  constexpr uint32_t XB = sizeof(XT);  // bytes per element of vec (4: operator ABI, 2: fused linear)
  // Clean slate for the compiler's wait-count model: the other roles sit upstream of this one in
  // the kernel's (static) control-flow graph, and whatever memory operation they leave "pending"
  // there (a FLAT access, a load into a register this role reuses) would otherwise be waited for
  // inside THIS role, conservatively.  Nothing is really outstanding here: the wait is free.
  __builtin_amdgcn_s_waitcnt(0);
  constexpr int L = F::kLut;
  constexpr int R = F::kRows;
  constexpr bool PAIR = BITS == 3 && HALF && SQLLM_PAIR3;          // 3-bit pair tables (batch tile 1)
  static_assert(!PAIR || (WAVES == 8 && BT == 1), "pair tables: wave w stages second index w");
  constexpr int ESTRIDE = (BITS == 4) ? 256 : 128;                 // bytes between consecutive entries
  constexpr int SUBB = (BITS == 4) ? (L * ESTRIDE) / 2 : L * ESTRIDE;  // LDS bytes per column sub-table
  // steps per chunk (4-bit: even, steps pair up for x; 3-bit: 12 VGPRs of weights per step)
  constexpr int NBUF = (BITS == 4) ? (BT <= 4 ? 4 : 2) : ((BT == 1 && !HALF) ? 2 : 1);
  constexpr int NXR = (BITS == 4) ? NBUF / 2 : 2 * NBUF;  // x registers per chunk and batch row
  constexpr int STEP = WAVES * 4;                // units a workgroup step covers
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, grp = lane >> 4;
  const int ct = bid % n_col_tiles;
  const int ks = bid / n_col_tiles;
  const int col0 = ct * kTileN;

  // ---- this workgroup's K range in units.  u_wave (this wave's first unit) is wave-uniform and
  //      drives every loop / guard; a lane's own unit is u_wave + grp (+ step offsets).
  const int u_beg = ks * units_per_wg;
  int u_end = u_beg + units_per_wg;
  if (u_end > units_total) u_end = units_total;
  const int u_last = u_end - 1;
  const int u_wave = u_beg + wave * 4;

  // Loads are UNCONDITIONAL with clamped addresses (a conditional load becomes a branch with an
  // immediate vmcnt(0)): lanes past N re-read the last valid 16 bytes of the row; steps past the
  // end of THIS workgroup's slice re-read the slice's own last unit (a cache hit -- clamping only
  // to the end of the matrix pulls other slices' rows from HBM: at 2-6 steps per wave that
  // over-fetch was 30-100 % of the useful traffic).  Such data is never accumulated.
  // Addresses are a wave-uniform base plus a 32-bit byte offset (one v_mul_u32_u24 + add per load;
  // 64-bit index arithmetic cost a v_mad_i64 and a 64-bit shift-add per load in an issue-bound
  // kernel).  The C ABI rejects matrices of 4 GiB or more.
  const int row_stride = N / 4;  // in 16-byte units
  int cidx = col0 / 4 + i16;
  if (cidx > row_stride - 1) cidx = row_stride - 1;
  const char* qbase = reinterpret_cast<const char*>(q);
  const uint32_t lane_bytes = 16u * (uint32_t)cidx;
  const uint32_t row_bytes = 16u * (uint32_t)row_stride;
  const uint32_t unit_bytes = (uint32_t)R * row_bytes;
  const char* xbase[BT];
#pragma unroll
  for (int b = 0; b < BT; ++b) xbase[b] = reinterpret_cast<const char*>(x + (size_t)(b0 + (b < nb ? b : nb - 1)) * K);

  // u = this wave's (uniform) unit for the chunk's first step; the lane's unit is u + grp
  auto load_chunk = [&](int u, u32x4 (&w)[NBUF][R], float (&xs)[NXR][BT]) {
#pragma unroll
    for (int s = 0; s < NBUF; ++s) {
      int uu = u + grp + s * STEP;
      if (uu > u_last) uu = u_last;
      const uint32_t off = __umul24((uint32_t)uu, unit_bytes) + lane_bytes;
#pragma unroll
      for (int r = 0; r < R; ++r)
        w[s][r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(qbase + (off + r * row_bytes)));
    }
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      if constexpr (BITS == 4) {
        // one x register serves two steps: lanes 0-7 of a row hold the first step's 8 k's,
        // lanes 8-15 the second step's
#pragma unroll
        for (int s2 = 0; s2 < NBUF / 2; ++s2) {
          int uu = u + grp + (2 * s2 + (i16 >> 3)) * STEP;
          if (uu > u_last) uu = u_last;
          const uint32_t off = XB * (8u * (uint32_t)uu + (i16 & 7));
          xs[s2][b] = (ABL & 16) ? 1.f + i16 : (float)*reinterpret_cast<const XT*>(xbase[b] + off);
        }
      } else {
#pragma unroll
        for (int s = 0; s < NBUF; ++s) {
          int uu = u + grp + s * STEP;
          if (uu > u_last) uu = u_last;
          const uint32_t off = XB * (32u * (uint32_t)uu + i16);
          xs[2 * s][b] = (float)*reinterpret_cast<const XT*>(xbase[b] + off);
          xs[2 * s + 1][b] = (float)*reinterpret_cast<const XT*>(xbase[b] + off + 16 * XB);
        }
      }
    }
  };

  // ---- codebook loads, then the first chunk's loads, go out before anything is waited for.
  // Staging is organised by LDS ROW (64 dwords = 256 B): a wave writes whole rows with lane ==
  // position in the row, so the writes of a half-wave hit 32 different banks.  (Letting thread t
  // write "its" float4 of the codebook block put 32 consecutive threads on 2 banks: 16-way
  // conflicts, measured as SQ_LDS_BANK_CONFLICT ~ SQ_ACTIVE_INST_LDS and 1-2 us per launch.)
  //   4-bit: row = (column pair, entry idx): 32 slots of the even column, 32 of the odd one
  //          (2 copies x 16 column groups each); wave w stages pair w % 2, entries
  //          [(w / 2) * EPW, + EPW), EPW = 32 / WAVES; lane >> 5 picks even / odd.
  //   3-bit: row = (column j, entry pair), 2 entries x 32 slots (2 copies x 16 column groups);
  //          wave w stages column j = w % 4, pairs [(w / 4) * RPW, + RPW), RPW = 16 / WAVES.
  constexpr int EPW = 32 / WAVES;                       // 4-bit: entries per wave
  constexpr int RPW = 16 / WAVES;                       // 3-bit: entry pairs per wave
  constexpr int NE = PAIR ? 9 : (BITS == 4) ? EPW : RPW;  // codebook values this thread stages
  float ev[NE];
  const int st_j = (BITS == 4) ? 2 * (wave & 1) + (lane >> 5) : (wave & 3);  // column this lane stages
  const int st_h = (BITS == 4) ? (wave >> 1) : (wave >> 2);
  if constexpr (!(ABL & 4)) {
    int c = col0 + 4 * i16 + (PAIR ? grp : st_j);
    if (c > N - 1) c = N - 1;
    const float* src = lut + (size_t)c * L;
    if constexpr (PAIR) {
      // thread = (slot i16, lane column grp, second index = wave): all 8 entries of its column,
      // plus the one that is the second element of every pair it writes
      const f32x4 ta = *reinterpret_cast<const f32x4*>(src), tb4 = *reinterpret_cast<const f32x4*>(src + 4);
      ev[0] = ta.x; ev[1] = ta.y; ev[2] = ta.z; ev[3] = ta.w;
      ev[4] = tb4.x; ev[5] = tb4.y; ev[6] = tb4.z; ev[7] = tb4.w;
      ev[8] = src[wave];
    } else if constexpr (BITS == 4) {
      static_assert(EPW % 4 == 0, "4-bit staging loads whole float4s");
#pragma unroll
      for (int i = 0; i < EPW / 4; ++i) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(src + st_h * EPW + 4 * i);
        ev[4 * i] = t.x; ev[4 * i + 1] = t.y; ev[4 * i + 2] = t.z; ev[4 * i + 3] = t.w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < RPW; ++i) ev[i] = src[2 * (st_h * RPW + i) + (lane >> 5)];
    }
  }
  // Fused linear: the top-X rows are folded into the dense tiles.  The first 64 column indices go
  // out first (consumed right after the staging barrier, while the weight loads behind them are
  // still in flight).  The load is UNCONDITIONAL -- without top-X rows it reads the codebook
  // pointer instead -- because a load under a branch makes the compiler wait for every
  // outstanding load at the join (measured: vmcnt(0) instead of vmcnt(6) before the codebook
  // staging, +0.3-0.6 us on every launch).  Operator launches keep the separate top-X role:
  // folding measured 6 % slower there (the matched workgroups become the tail of the launch).
#ifdef SQLLM_ABLATION_BUILD
  // timeline probe (measurement build): sg.bias, unused by operator launches, carries a buffer of
  // 8 x u64 per workgroup; wave 0 stamps entry / barrier passed / decode done, the combining wave
  // stamps the end.  s_memrealtime = 100 MHz constant clock, comparable across CUs.
  unsigned long long* tl = (!lin && sg.bias) ? reinterpret_cast<unsigned long long*>(const_cast<float*>(sg.bias)) +
                                                   8ull * (blockIdx.x + (unsigned long long)gridDim.x * blockIdx.y) : nullptr;
  if (tl && tid == 0) tl[0] = __builtin_amdgcn_s_memrealtime();
#endif
  // Branches first: between the loads below and the codebook staging there must be NO control
  // flow, or the staging waits for every outstanding load (vmcnt(0)) instead of its own.
  constexpr int kCodebookFloats = PAIR ? 4 * 64 * 128 / 4 : 4 * SUBB / 4;  // the four column sub-tables
  float* topx_sum = lds + kCodebookFloats + WAVES * BT * kTileN + 4;  // [BT][64], fused linear only
  // epilogue ticket (the dword after the slabs) and, 4 dwords on, the BT * 64 top-X sums
  for (int i = tid; i < 4 + BT * kTileN; i += WAVES * 64) lds[kCodebookFloats + WAVES * BT * kTileN + i] = 0.f;
  constexpr bool FOLD = !std::is_same<XT, float>::value;  // == fused-linear instantiation
  const bool fold_topx = FOLD && sg.full_rows != nullptr;
  int topx_idx = -1;
  if constexpr (FOLD) {
    const int* fi = fold_topx ? sg.full_idx : reinterpret_cast<const int*>(lut);
    topx_idx = fi[fold_topx ? (lane < sg.gm.topX ? lane : sg.gm.topX - 1) : 0];
  }
  u32x4 w0[NBUF][R];
  float x0[NXR][BT];
  load_chunk(u_wave, w0, x0);
  // keep these loads ABOVE the staging barrier (LLVM would otherwise sink them below it)
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);

  // ---- stage the codebooks (row-wise, see above) ----
  if constexpr (!(ABL & 4)) {
    if constexpr (PAIR) {
      // sub-table of lane column j: 64 entry rows of 128 B (16 slots x 8 B); entry i0 + 8 * wave
      char* dst = reinterpret_cast<char*>(lds) + grp * 8192 + wave * 8 * 128 + i16 * 8;
#pragma unroll
      for (int i0 = 0; i0 < 8; ++i0) *reinterpret_cast<f32x2*>(dst + i0 * 128) = f32x2{ev[i0], ev[8]};
    } else if constexpr (BITS == 4) {
      // row (pair, idx) starts at pair * 4096 + idx * 256; this lane's dword in it is `lane`
      float* dst = lds + ((wave & 1) * 4096 + st_h * EPW * ESTRIDE) / 4 + lane;
#pragma unroll
      for (int i = 0; i < EPW; ++i) dst[i * (ESTRIDE / 4)] = ev[i];
    } else {
      // entry idx = 2 * pair + (lane >> 5); slot = lane & 31
      float* dst = lds + (st_j * SUBB) / 4 + (lane >> 5) * (ESTRIDE / 4) + (lane & 31);
#pragma unroll
      for (int i = 0; i < RPW; ++i) dst[2 * (st_h * RPW + i) * (ESTRIDE / 4)] = ev[i];
    }
  }

  f32x2 acc[2][BT];  // [column pair][batch row]: columns 2p and 2p+1 of the lane's four
#pragma unroll
  for (int jp = 0; jp < 2; ++jp)
#pragma unroll
    for (int b = 0; b < BT; ++b) acc[jp][b] = f32x2{0.f, 0.f};

  // per-lane LDS byte offset inside an entry row (4-bit) / per-sub-table bases (3-bit)
  const uint32_t lane_off = 4 * (i16 + 16 * (grp & 1));  // 4-bit: dword slot inside a 128-byte half row
  uint32_t tb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) tb[j] = PAIR ? j * 8192 + 8 * i16 + (SQLLM_PAIR3_NOCONFLICT ? 128 * (grp & 1) : 0) : j * SUBB + 4 * (i16 + 16 * (grp & 1));
  f32x2 accp[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};  // PAIR: (even k, odd k) per column

  __syncthreads();  // codebooks visible
#ifdef SQLLM_ABLATION_BUILD
  if (tl && tid == 0) tl[1] = __builtin_amdgcn_s_memrealtime();
#endif

  // u = this wave's (uniform) unit for the chunk's first step: guards are scalar branches; only the
  // per-row validity of a slice's ragged end is per lane (it zeroes x, no divergence)
  auto decode_chunk = [&](int u, const u32x4 (&w)[NBUF][R], const float (&xs)[NXR][BT]) {
    if constexpr (BITS == 4) {
#pragma unroll
      for (int s2 = 0; s2 < NBUF / 2; ++s2) {
        const int ua = u + 2 * s2 * STEP, ub = ua + STEP;
        if constexpr (HALF) {
          if (ua < u_end) step4_half<BT, 0, ABL>(w[2 * s2][0], xs[s2], ua + grp < u_end, lane_off, acc);
          if (ub < u_end) step4_half<BT, 8, ABL>(w[2 * s2 + 1][0], xs[s2], ub + grp < u_end, lane_off, acc);
        } else {
          if (ua < u_end) step4<BT, 0, ABL>(w[2 * s2][0], xs[s2], ua + grp < u_end, lane_off, acc);
          if (ub < u_end) step4<BT, 8, ABL>(w[2 * s2 + 1][0], xs[s2], ub + grp < u_end, lane_off, acc);
        }
      }
    } else {
#pragma unroll
      for (int s = 0; s < NBUF; ++s) {
        const int ua = u + s * STEP;
        if constexpr (PAIR) {
          if (ua < u_end) step3_pair(w[s], xs[2 * s][0], xs[2 * s + 1][0], ua + grp < u_end, tb, accp);
        } else {
          if (ua < u_end) step3<BT, ABL, HALF>(w[s], xs[2 * s], xs[2 * s + 1], ua + grp < u_end, tb, acc);
        }
      }
    }
  };

  // ---- folded top-X: rows whose column lies in this tile are this workgroup's job too -- their
  // dot product over this K slice joins the column's partial sum, so they need no role, no
  // atomics and no counting of their own.  Every wave scans the indices itself (no barrier); a
  // match is rare (topX columns out of N), and its loads overlap the first chunk's.
  if (fold_topx) {
    const int topX = sg.gm.topX;
    const int k_beg = u_beg * F::kK, k_end = u_end * F::kK;
    for (int j0 = 0; j0 < topX; j0 += 64) {
      const int cj = (j0 == 0) ? topx_idx : sg.full_idx[j0 + lane < topX ? j0 + lane : topX - 1];
      unsigned long long m = __ballot(j0 + lane < topX && cj >= col0 && cj < col0 + kTileN);
      while (m) {
        const int jl = __builtin_ctzll(m);
        m &= m - 1;
        const int j = j0 + jl;
        const int cc = __builtin_amdgcn_readlane(cj, jl) - col0;
#pragma unroll
        for (int b = 0; b < BT; ++b) {
          float p = 0.f;
          for (int k = k_beg + wave * 64 + lane; k < k_end; k += WAVES * 64)
            p = __builtin_fmaf(sg.full_rows[(size_t)k * topX + j],
                               (float)x[(size_t)(b0 + (b < nb ? b : nb - 1)) * K + k], p);
          p = wave_sum(p);
          if (lane == 0) atomicAdd(topx_sum + b * kTileN + cc, p);
        }
      }
    }
  }

  decode_chunk(u_wave, w0, x0);
  for (int u0 = u_wave + NBUF * STEP; u0 < u_end; u0 += NBUF * STEP) {  // scalar loop
    u32x4 w[NBUF][R];
    float xs[NXR][BT];
    load_chunk(u0, w, xs);
    __builtin_amdgcn_sched_barrier(0);
    decode_chunk(u0, w, xs);
  }

#ifdef SQLLM_ABLATION_BUILD
  if (tl && tid == 0) tl[2] = __builtin_amdgcn_s_memrealtime();
  if (tl && lane == 0) tl[4 + (wave & 3)] = __builtin_amdgcn_s_memrealtime();  // decode end of waves 0-3
#endif
  if constexpr (PAIR) {
    acc[0][0] = f32x2{accp[0].x + accp[0].y, accp[1].x + accp[1].y};
    acc[1][0] = f32x2{accp[2].x + accp[2].y, accp[3].x + accp[3].y};
  }
  dense_epilogue<BT, WAVES, ABL>(acc, lds + kCodebookFloats, topx_sum, fold_topx, y, N, col0, b0, nb, lane, wave, sg, lin
#ifdef SQLLM_ABLATION_BUILD
                                 , tl
#endif
  );
}

// ------------------------------------------------------------------------------------------------
// CSR role: one workgroup per chunk of kCsrChunk consecutive non-zeros (balanced by nnz, so a few
// very long rows cost nothing extra -- the reference walks one row per thread serially,
// quant_cuda_kernel.cu:1049-1058).
//
// The role is latency-bound (a chunk is 8 KiB of cols/vals), so it is organised as TWO rounds of
// independent global loads and nothing else dependent on memory:
//   round 1: this thread's cols/vals (coalesced) + ONE sampled probe of `rows` per thread
//            (rows[t * S], S = ceil((N+1)/T)); two block-wide counts turn the probes into the
//            sample intervals that contain the chunk's first and last non-zero;
//   round 2: the x gather (needs cols) + the row pointers of every row between those two
//            intervals, staged straight into LDS (needs the counts);
//   then, LDS only: each non-zero finds its row by binary search in the staged pointers, products
//   are summed per row in LDS, and each touched row leaves as one atomic.
// ------------------------------------------------------------------------------------------------
template <int T, int BT, typename XT, typename AT, bool XTMODE = false>
__device__ __forceinline__ void csr_role(const XT* x, AT* __restrict__ y,
                                         const int* __restrict__ rows, const int* __restrict__ cols,
                                         const float* __restrict__ vals, int nnz, int K, int N, int b0,
                                         int nb, int chunk, float* lds, const Segment* lin, int lin_or_abl_bits = 0,
                                         const float* __restrict__ xT = nullptr, int Bp = 0) {
  constexpr bool LIN = sizeof(AT) == 8;
  const int tid = threadIdx.x;
  const int e0 = chunk * kCsrChunk;
  int e1 = e0 + kCsrChunk;
  if (e1 > nnz) e1 = nnz;
  if (e0 >= e1) return;
#ifdef SQLLM_ABLATION_BUILD
  const int cabl = lin_or_abl_bits;  // 1 = skip the role, 2 = skip the flush, 4 = skip the accumulation
  if (cabl & 1) return;
#endif

  // ---- round 1 ----
  constexpr int EPT = kCsrChunk / T;  // non-zeros per thread: e0 + tid + T * i (coalesced)
  int col[EPT];
  float val[EPT];
  // element of (thread, i): interleaved over the workgroup, or -- transposed-vec mode -- EPT runs of 64
  // that are consecutive within a wave (so that only a wave's first and last row are shared with its
  // neighbours)
  auto elem = [&](int i) { return XTMODE ? e0 + (tid >> 6) * (64 * EPT) + 64 * i + (tid & 63) : e0 + tid + T * i; };
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    int e = elem(i);
    if (e > e1 - 1) e = e1 - 1;  // clamped re-read; masked below
    col[i] = cols[e];
    val[i] = vals[e];
  }
  const int S = (N + T) / T;  // sample stride: T samples cover rows[0 .. N]
  const int si = tid * S;
  const int probe = rows[si < N ? si : N];
  // rows is non-decreasing with rows[0] = 0, so both predicates are true for a prefix of samples
  const int cnt_lo = __syncthreads_count(si <= N && probe <= e0);
  const int cnt_hi = __syncthreads_count(si <= N && probe <= e1 - 1);
  const int c_lo = (cnt_lo > 0 ? cnt_lo - 1 : 0) * S;  // rows[c_lo] <= e0
  int c_hi = cnt_hi * S;                                // rows[c_hi] > e1 - 1 (or the end)
  if (c_hi > N) c_hi = N;
  const int n = c_hi - c_lo + 1;  // staged row pointers rows[c_lo .. c_hi]; candidate rows: n - 1
  const bool in_lds = n <= kCsrSpanMax;

  // ---- round 2 ----
  int* srows = reinterpret_cast<int*>(lds);  // [kCsrSpanMax]
  float* sacc = lds + kCsrSpanMax;           // [kCsrSpanMax]
  // the gather goes out first: the staging loop below waits for its own loads before it stores
  float xg[EPT];
#pragma unroll
  for (int i = 0; i < EPT; ++i) xg[i] = XTMODE ? 0.f : (float)x[(size_t)b0 * K + col[i]];  // first batch row's gather
  // Batch rows go through in groups of `g`: as many as have room for their n row sums each in the
  // LDS accumulator (all of them for typical chunks, which span 50-100 rows), so a batched op
  // pays the zero / accumulate / flush round and its barriers once, not once per row, and the x
  // gathers of all rows are in flight together.
  int g = 1;
  if (in_lds) {
    g = kCsrSpanMax / n;
    if (g > nb) g = nb;
    if (g < 1) g = 1;
    for (int i = tid; i < n; i += T) srows[i] = rows[c_lo + i];
    for (int i = tid; i < n * g; i += T) sacc[i] = 0.f;  // first group's sums (no barrier of its own)
  }
  // transposed-vec mode: sums of the pass's rows, tile[row][local column] (odd stride: the lanes of a
  // wave -- one row each -- write one bank each), zeroed here, flushed coalesced along the columns
  const bool use_tile = XTMODE && in_lds && n <= kCsrXtSpan;
  const int TS = n | 1;
  float* tile = lds + kCsrSpanMax;
  if (use_tile) for (int i = tid; i < 64 * TS; i += T) tile[i] = 0.f;
  __syncthreads();

  // local row of each non-zero: largest i with rows[c_lo + i] <= e
  int lr[EPT];
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    const int e = elem(i);
    int lo = 0, hi = n - 1;  // answer in [lo, hi): rows[c_lo + n - 1] > e by construction
    if (hi < 1) hi = 1;
    if (in_lds) {
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (srows[mid] <= e) lo = mid; else hi = mid;
      }
    } else {  // a chunk spanning > kCsrSpanMax rows (extremely sparse region): search in global memory
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (rows[c_lo + mid] <= e) lo = mid; else hi = mid;
      }
    }
    lr[i] = (e < e1) ? lo : -1;
  }
  // segment structure of each 64-lane run of non-zeros (fixed for all batch rows): bit d = the lane
  // 2^d below belongs to the same row (take its partial sum in scan step d), bit 6 = last lane of
  // its row segment
  unsigned seg[EPT];
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    const int lane = tid & 63;
    unsigned m = 0;
#pragma unroll
    for (int d = 0; d < 6; ++d) {
      const int below = __shfl_up(lr[i], 1 << d, 64);
      if (lane >= (1 << d) && below == lr[i]) m |= 1u << d;
    }
    const int above = __shfl_down(lr[i], 1, 64);
    if (lane == 63 || above != lr[i]) m |= 64u;
    seg[i] = m;
  }

  if constexpr (XTMODE) {
    // Wide batches with a TRANSPOSED copy of vec (xT[k][row], written by sqllm_transpose_vec just
    // before this launch): lane = batch row.  A wave walks its 64 * EPT consecutive non-zeros one
    // at a time -- column, value and row come out of the owning lane with v_readlane, so control flow
    // and addresses are scalar -- and every lane loads ITS row's element of xT[k] (one coalesced
    // read per non-zero instead of one gather per row, 4 K bytes apart) and multiplies.  At the last
    // non-zero of a row the lanes park their sums in tile[row][column]: a plain store, or an LDS add
    // for the wave's first and last row (which the neighbouring waves may hold parts of).  The tile
    // leaves with the lanes along the COLUMNS: coalesced atomics (lanes along the rows would hit
    // one cache line each: measured 2.1 ms of a 4.6 ms launch at 2048 rows).
    const int lane = tid & 63;
    const bool row_ok = lane < nb;
    const float* xl = xT + (b0 + (row_ok ? lane : 0));
    unsigned long long ends[EPT], valid[EPT];
    int n_valid = 0;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      ends[i] = __ballot((seg[i] & 64u) && lr[i] >= 0);
      valid[i] = __ballot(lr[i] >= 0);
      n_valid += __builtin_popcountll(valid[i]);
    }
#pragma unroll
    for (int i = 0; i + 1 < EPT; ++i)  // a row that runs on into the next run keeps its sum in the register
      if ((valid[i + 1] & 1ull) && __builtin_amdgcn_readlane(lr[i], 63) == __builtin_amdgcn_readlane(lr[i + 1], 0)) ends[i] &= ~(1ull << 63);
    float acc = 0.f;
    bool first_seg = true;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      constexpr int U = 32;  // loads in flight per wave: the role is latency-bound (a chunk is 1-2 workgroups per CU)
      for (int j0 = 0; j0 < 64; j0 += U) {
        if (((valid[i] >> j0) & 1ull) == 0) break;  // (valid lanes are a prefix)
        float xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int k = __builtin_amdgcn_readlane(col[i], j0 + u);
          xv[u] = xl[(size_t)k * Bp];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int j = j0 + u;
          const float v = ((valid[i] >> j) & 1ull) ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, val[i]), j)) : 0.f;
          acc = __builtin_fmaf(v, xv[u], acc);
          if ((ends[i] >> j) & 1ull) {
            const int r = __builtin_amdgcn_readlane(lr[i], j);
            if (use_tile) {
              float* slot = tile + lane * TS + r;
              if (first_seg || 64 * i + j == n_valid - 1) atomicAdd(slot, acc);  // LDS float atomic: lane by lane, twice per wave
              else *slot = acc;
            } else if (row_ok) {
              acc_add(reinterpret_cast<float*>(y) + (size_t)(b0 + lane) * N + c_lo + r, acc);
            }
            first_seg = false;
            acc = 0.f;
          }
        }
      }
    }
    if (use_tile) {
      __syncthreads();
      const int nm1 = n - 1;
      for (int idx = tid; idx < nm1 * nb; idx += T) {
        const int b = idx / nm1, r = idx - b * nm1;
        const float sum = tile[b * TS + r];
        if (sum != 0.f) acc_add(reinterpret_cast<float*>(y) + (size_t)(b0 + b) * N + c_lo + r, sum);
      }
    }
  } else {
  const int nm1 = n - 1 > 0 ? n - 1 : 1;
  for (int bs = 0; bs < nb; bs += g) {
    const int gb = nb - bs < g ? nb - bs : g;
    if (in_lds && bs > 0) {
      for (int i = tid; i < n * gb; i += T) sacc[i] = 0.f;
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      float xv[BT];
#pragma unroll
      for (int bb = 0; bb < BT; ++bb) {  // unconditional loads (rows past the group re-read its last row)
        const int bi = bs + (bb < gb ? bb : gb - 1);
        xv[bb] = (float)x[(size_t)(b0 + bi) * K + col[i]];
      }
      if (bs == 0) xv[0] = xg[i];
      // a wave holds 64 consecutive non-zeros, i.e. a few whole or partial rows: segmented
      // inclusive scan by row across the lanes, then ONE add per row segment (from its last lane)
      // instead of 64 adds that collide on 2-3 addresses -- LDS float atomics to one address are
      // executed one lane at a time (measured: 17 of 49 us of a batch-8 13B hybrid launch).
      const int r = lr[i];
      const unsigned sm = seg[i];
#ifdef SQLLM_ABLATION_BUILD
      if (cabl & 4) continue;
#endif
#pragma unroll
      for (int bb = 0; bb < BT; ++bb) {
        if (bb < gb) {
          float p = val[i] * xv[bb];
#pragma unroll
          for (int d = 0; d < 6; ++d) {
            const float up = __shfl_up(p, 1 << d, 64);
            if (sm & (1u << d)) p += up;
          }
          if ((sm & 64u) && r >= 0) {
            if (in_lds) atomicAdd(sacc + bb * n + r, p);
            else acc_add(y + (size_t)(b0 + bs + bb) * N + c_lo + r, p);
          }
        }
      }
    }
    if (in_lds) {
      __syncthreads();
#ifdef SQLLM_ABLATION_BUILD
      if (cabl & 2) continue;
#endif
      for (int idx = tid; idx < nm1 * gb && n > 1; idx += T) {
        const int bb = idx / nm1;
        const int i = idx - bb * nm1;
        const float sum = sacc[bb * n + i];
        const size_t at = (size_t)(b0 + bs + bb) * N + c_lo + i;
        if constexpr (LIN) {
          // one COUNTED contribution per row this chunk holds a part of, whatever its value
          const int r0 = srows[i], r1 = srows[i + 1];
          if ((r0 > e0 ? r0 : e0) < (r1 < e1 ? r1 : e1)) {
            const u64 mine = kCountUnit + to_fixed(sum);
            const unsigned target = (unsigned)lin->gm.k_slices + (unsigned)csr_chunks_of_row(r0, r1);
            column_done(*lin, y + at, atomicAdd(y + at, mine) + mine, target, at, c_lo + i);
          }
        } else {
          if (sum != 0.f) acc_add(y + at, sum);
        }
      }
      __syncthreads();
    } else if constexpr (LIN) {
      // (g == 1 here) the values went in uncounted, one add per non-zero; once they are
      // acknowledged, count this chunk on every row it holds a part of
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      for (int i = tid; i < n - 1; i += T) {
        const int r0 = rows[c_lo + i], r1 = rows[c_lo + i + 1];
        if ((r0 > e0 ? r0 : e0) < (r1 < e1 ? r1 : e1)) {
          const size_t at = (size_t)(b0 + bs) * N + c_lo + i;
          const unsigned target = (unsigned)lin->gm.k_slices + (unsigned)csr_chunks_of_row(r0, r1);
          column_done(*lin, y + at, atomicAdd(y + at, kCountUnit) + kCountUnit, target, at, c_lo + i);
        }
      }
    }
  }
  }  // !XTMODE
}

// ------------------------------------------------------------------------------------------------
// top-X role: full_rows is fp32 [K, topX] row-major; a workgroup takes kTopxRows consecutive k's,
// i.e. one contiguous slab of kTopxRows*topX floats, and streams it coalesced (the reference keeps
// topX of 128 lanes busy with stride-topX reads, quant_cuda_kernel.cu:1113-1118).
// ------------------------------------------------------------------------------------------------
template <int T, typename XT, typename AT>
__device__ __forceinline__ void topx_role(const XT* x, AT* __restrict__ y,
                                          const float* __restrict__ full_rows,
                                          const int* __restrict__ full_idx, int topX, int K, int N,
                                          int b0, int nb, int slab, float* lds) {
  const int tid = threadIdx.x;
  const int k0 = slab * kTopxRows;
  int k1 = k0 + kTopxRows;
  if (k1 > K) k1 = K;
  const int nel = (k1 - k0) * topX;
  const float* fr = full_rows + (size_t)k0 * topX;
  if (topX <= 16) {
    // The usual case (the reference uses topX = 10).  Lane l of a 16-lane row owns column l (lanes
    // >= topX idle) and the 32 lane rows of the workgroup take k0 + row, + 32, + 64, + 96: a wave
    // reads 4 consecutive rows of the slab (contiguous), every thread keeps ONE partial sum in a
    // register, two cross-lane adds fold the wave's 4 lane rows, the 8 waves meet in LDS through
    // plain stores.  No LDS atomics (64 lanes on 10 addresses execute one lane at a time: that and
    // two more barriers cost 0.6-0.9 us on the grouped 7B launches), one barrier per batch row.
    static_assert(T == 512, "32 lane rows x 4 k's cover the 128-k slab");
    const int c = tid & 15, krow = tid >> 4;  // krow 0..31
    const int lane = tid & 63, wave = tid >> 6;
    const bool live = c < topX;
    float frv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int k = k0 + krow + 32 * i;
      if (k > k1 - 1) k = k1 - 1;  // clamped re-read, masked below
      frv[i] = live ? full_rows[(size_t)k * topX + c] : 0.f;
    }
    const int dst = live ? full_idx[c] : 0;
    for (int b = 0; b < nb; ++b) {
      const XT* xb = x + (size_t)(b0 + b) * K;
      float p = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = k0 + krow + 32 * i;
        p = __builtin_fmaf(frv[i], k < k1 ? (float)xb[k] : 0.f, p);
      }
      p += __shfl_xor(p, 16, 64);
      p += __shfl_xor(p, 32, 64);
      if (b > 0) __syncthreads();  // the previous batch row's sums have been read
      if (lane < 16) lds[wave * 16 + lane] = p;
      __syncthreads();
      if (tid < topX) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < T / 64; ++w) sum += lds[w * 16 + tid];
        acc_add(y + (size_t)(b0 + b) * N + dst, sum);
      }
    }
    return;
  }
  const bool in_lds = topX <= kTopxLds;
  float* sacc = lds;
  for (int b = 0; b < nb; ++b) {
    const XT* xb = x + (size_t)(b0 + b) * K + k0;
    AT* yb = y + (size_t)(b0 + b) * N;
    if (in_lds) {
      for (int c = tid; c < topX; c += T) sacc[c] = 0.f;
      __syncthreads();
    }
    for (int e = tid; e < nel; e += T) {
      const int kk = e / topX;
      const int c = e - kk * topX;
      const float p = fr[e] * (float)xb[kk];
      if (in_lds) atomicAdd(sacc + c, p); else acc_add(yb + full_idx[c], p);
    }
    if (in_lds) {
      __syncthreads();
      for (int c = tid; c < topX; c += T) acc_add(yb + full_idx[c], sacc[c]);
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------------
// the fused kernel
// ------------------------------------------------------------------------------------------------
// Occupancy is what this kernel lives on (measured, DESIGN.md 4.1: thread-level parallelism beats
// instruction-level parallelism here -- software-pipelining the stages at 104 VGPRs lost 8 %, while
// halving the live lookups won up to 18 %): the decode stages work on one column pair at a time
// (16 live lookups instead of 32), which lets the batch-1 kernels fit 64 VGPRs, i.e. FOUR 8-wave
// workgroups per CU; the wider batch tiles take what they need up to 128 (two per CU).
template <int BITS, int BT, int WAVES, int ABL, bool LIN>
__global__ void __launch_bounds__(WAVES * 64, (ABL & 64) ? 8 : ((BT == 1 && SQLLM_HALF_STAGES) ? 8 : 4))
sqllm_fused_matvec(const void* xv, const GroupArgs ga) {
  constexpr bool HALF = SQLLM_HALF_STAGES && BT == 1;  // wider batch tiles: the per-row x broadcasts would be live twice
  constexpr int T = WAVES * 64;
  constexpr int kLds = lds_floats(Fmt<BITS>::kLut, WAVES, BT, BITS == 3 && BT == 1 && SQLLM_HALF_STAGES && SQLLM_PAIR3);
  __shared__ __attribute__((aligned(16))) float lds[kLds];
  using XT = typename XType<LIN>::type;
  using AT = typename AccType<LIN>::type;
  const XT* x = reinterpret_cast<const XT*>(xv);

  // The argument block lives in memory and is read with scalar loads; a DEPENDENT scalar load costs
  // 0.15-0.17 us here (tools/experiments/dispatch_ramp.hip: a chain of 8 takes 1.25 us, for the first
  // workgroup of a CU and for the later ones alike).  Reading fields where they are used made the
  // prologue a chain of 6-8 such loads (block0 -> s -> sparse_last -> dense_block0 -> dense_blocks ->
  // ... -> q, lut) in front of the first vector load of every workgroup.  So: ONE round of loads
  // fetches vec's address, the block table and -- speculatively -- the whole of segment 0 into
  // registers; a workgroup of another segment pays a second round for its own descriptor.
  Segment sg = ga.seg[0];
  const int n_seg = ga.n_seg, blk1 = ga.block0[1], blk2 = ga.block0[2], blk3 = ga.block0[3];
  asm volatile("" ::SQLLM_SEG_OPERANDS(sg), "s"(x), "s"(n_seg), "s"(blk1), "s"(blk2), "s"(blk3));
  __builtin_amdgcn_sched_barrier(0);  // (or the scheduler starts on the block table after the first few loads, waits, and issues the rest behind that wait)
  // which op of the launch this workgroup belongs to (wave-uniform; 1 segment = a plain op)
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
  const int b0 = blockIdx.y * BT;
  int nb = gm.batch - b0;
  if (nb > BT) nb = BT;

  // role by block id within the segment: [sparse | pad | dense] or, with sparse_last, [dense | sparse]
  int d, sp;
  if (gm.sparse_last & 1) {
    d = bid;
    sp = bid - gm.dense_blocks;
  } else {
    d = bid - gm.dense_block0;
    sp = bid < gm.dense_block0 ? bid : -1;
  }
  if (d >= 0 && d < gm.dense_blocks) {
    dense_role<BITS, BT, WAVES, ABL, XT, HALF>(x, reinterpret_cast<const u32x4*>(sg.q), sg.y, sg.lut, gm.K, gm.N, b0, nb,
                                         d, gm.col_tiles, gm.units_total, gm.units_per_wg, lds, sg, LIN ? &ga.seg[s] : nullptr);
  } else if (sp >= 0 && sp < gm.csr_blocks) {
    csr_role<T, BT, XT, AT>(x, reinterpret_cast<AT*>(sg.y), sg.rows, sg.cols, sg.vals, gm.nnz, gm.K, gm.N, b0, nb, sp, lds,
                        LIN ? &ga.seg[s] : nullptr, gm.sparse_last >> 1);
  } else if (sp >= gm.csr_blocks && sp < gm.csr_blocks + gm.topx_blocks) {
    // (never taken when the plan folds the top-X rows into the dense tiles)
    topx_role<T, XT, AT>(x, reinterpret_cast<AT*>(sg.y), sg.full_rows, sg.full_idx, gm.topX, gm.K, gm.N, b0, nb, sp - gm.csr_blocks, lds);
  }
}

// ------------------------------------------------------------------------------------------------
// Wide-batch dense role: fp32 MATRIX cores (the *_batched operators from `mfma_min_batch` rows up).
//
// The batch tiles above reuse one lookup for up to 8 vector FMAs, so a batched op costs VALU time
// proportional to the batch (13B gate/up at batch 8: 2.8 x the batch-1 launch) and re-streams the
// weights once per 8 rows.  Here the dequantised weights are the B operand of
// v_mfma_f32_16x16x4_f32 (exact fp32: a plain fma chain; 32 cycles per SIMD, the fp32 vector rate):
//     lane (c = l % 16, kq = l / 16)  loads the usual 16 bytes: 4 adjacent columns 4c .. 4c+3 of ITS
//     qweight row r0 + kq (so one wave load is 4 rows x 64 columns, as in the other kernels) and
//     supplies, in step s = 0..7 and column block j = 0..3,
//         B = W[8 (r0 + kq) + s, col0 + 4c + j]     -- nibble s of its own word: no cross-lane traffic
//         A = vec[m0 + 16 mb + c, 8 (r0 + kq) + s]  -- batch row c of row block mb, the same k
//     and accumulates D[v] = mul[m0 + 16 mb + 4 kq + v, col0 + 4c + j].
// (The matrix instruction only needs A and B to agree on WHICH four k's a step multiplies; taking
// "nibble s of four consecutive rows" instead of four consecutive k's is what keeps every lane on
// its own loaded word.)  3-bit: the same with 32-k units and four phases of 8 k's per unit.
// A weight is looked up once however many batch rows there are; MB blocks of 16 rows (<= 64 rows
// per pass) cost MB matrix instructions per step.  Budget per 4 rows x 64 columns of weights:
// 44 VALU + 32 lookups against MB x 1024 matrix-pipe cycles -- the op is matrix-bound from the
// first block on (~2.1 TB/s of 4-bit weights per 16 rows at a fully busy matrix pipe), so every
// batch up to 16 costs the same.  Measured (13B gate/up shape, 5120 x 13824, 4-bit, MI355X):
// 30 us for 1..16 rows with the matrix pipe 50 % busy (rocprofv3 SQ_VALU_MFMA_BUSY_CYCLES: all
// workgroups are resident at once and run their prologue / matrix / epilogue phases in step),
// 47-51 TFLOP/s from 32 rows up -- against 38 us (8 rows), 74 us (16 rows) and 37 TFLOP/s for the
// 8-row batch tiles (which also serve q/k/v and gate/up as ONE launch); hence the default switch-over
// at 9 rows, where the tiles would need a second pass.
// Batches beyond 64 rows put the next 64 rows in the next blockIdx.y: the weights of a workgroup's
// slice (<= 256 KB) are then re-read from L2 / Infinity Cache, not from HBM.
// vec reaches the A operands through a PRIVATE LDS tile per wave (the waves of a workgroup work on
// different k's, so there is no barrier): 16 MB rows x 32 k's, written as 16-byte pieces by the lanes
// that loaded them one chunk ahead, read back as two ds_read_b128 per row block.
// Waves split the K slice, meet in LDS (fp32 adds) and leave as one atomic per (row, column).
// The CSR and top-X roles are the ones of the other kernels, run over the pass's rows 8 at a time.
// ------------------------------------------------------------------------------------------------
constexpr int kXtStride = 36;  // floats per row of a wave's x tile (32 k's + 4: rows 4 apart in banks)
constexpr int mfma_codebook_floats(int bits) { return bits == 4 ? 2 * 4096 / 4 : 4 * 8 * 128 / 4; }
constexpr int mfma_lds_floats(int bits, int mb, int waves) {
  // codebooks, then the waves' x tiles; the epilogue's slabs [waves][16][64] reuse the tile area
  return mfma_codebook_floats(bits) + cmax(waves * 16 * mb * kXtStride, waves * 16 * 64);
}

template <int BITS, int MB, int WAVES>
__device__ __forceinline__ void dense_role_mfma(const float* __restrict__ x, const u32x4* __restrict__ q,
                                                float* __restrict__ y, const float* __restrict__ lut, int K, int N,
                                                int batch, int m0, int bid, int n_col_tiles, int units_total,
                                                int units_per_wg, float* lds) {
  using F = Fmt<BITS>;
  constexpr int L = F::kLut, R = F::kRows, KU = F::kK;
  constexpr int NPH = KU / 8;          // phases of 8 k's per unit (4-bit: 1, 3-bit: 4)
  constexpr int TR = 16 * MB;          // rows of the x tile
  constexpr int XL = TR / 8;           // 16-byte pieces of vec a lane loads per phase
  constexpr int ESTRIDE = (BITS == 4) ? 256 : 128;
  constexpr int SUBB = (BITS == 4) ? (L * ESTRIDE) / 2 : L * ESTRIDE;
  __builtin_amdgcn_s_waitcnt(0);  // clean slate for the compiler's wait-count model (see dense_role)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, grp = lane >> 4;
  if constexpr ((SQLLM_MFMA_VAR & 64) != 0) {
    // measurement: STATIC, DIFFERENT priorities for the waves that share a SIMD (waves w and w + 4 of a
    // workgroup; with bit 128 also the co-resident workgroup, which is 256 ids away), so that they
    // fall out of step: one decodes while the other holds the matrix pipe
    const int pr = (wave >> 2) + ((SQLLM_MFMA_VAR & 128) ? 2 * ((bid >> 8) & 1) : 0);
    if (pr == 1) __builtin_amdgcn_s_setprio(1);
    else if (pr == 2) __builtin_amdgcn_s_setprio(2);
    else if (pr == 3) __builtin_amdgcn_s_setprio(3);
  }
  // LDS: codebooks at address 0 in the layout of dense_role (conflict-free lookups), then the x tiles
  // (one per wave); the epilogue's slabs live where the tiles were
  constexpr int kCb = mfma_codebook_floats(BITS);
  float* slabs = lds + kCb;
  float* xt = lds + kCb + wave * (TR * kXtStride);
  constexpr int EPW = 32 / WAVES, RPW = 16 / WAVES;
  constexpr int NE = (BITS == 4) ? EPW : RPW;
  const int st_j = (BITS == 4) ? 2 * (wave & 1) + (lane >> 5) : (wave & 3);
  const int st_h = (BITS == 4) ? (wave >> 1) : (wave >> 2);
  const int row_stride = N / 4;  // in 16-byte units
  const char* qbase = reinterpret_cast<const char*>(q);
  const uint32_t row_bytes = 16u * (uint32_t)row_stride;
  // vec pieces: lane -> (tile row l / 8 + 8 j, lane row piece >> 1 of the group, half piece & 1)
  int xrow[XL];
#pragma unroll
  for (int j = 0; j < XL; ++j) {
    int r = m0 + (lane >> 3) + 8 * j;
    if (r > batch - 1) r = batch - 1;  // rows past the batch re-read its last row; never stored
    xrow[j] = r * K + 4 * (lane & 1);
  }
  const int xkq = (lane >> 1) & 3;  // which lane row's k's this lane's pieces belong to
  const uint32_t lane_off = 4 * (i16 + 16 * (grp & 1));
  uint32_t tb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) tb[j] = j * SUBB + 4 * (i16 + 16 * (grp & 1));
  float* xt_w = xt + (lane >> 3) * kXtStride + 8 * xkq + 4 * (lane & 1);  // where this lane parks its pieces
  const float* xt_r = xt + i16 * kXtStride + 8 * grp;                     // batch row i16 of block 0, this lane row's 8 k's

  // ---- this workgroup's share: units [bid * units_per_wg, + units_per_wg) of the FLATTENED
  // (column tile, unit) space -- every workgroup the same number of units whatever N and K are, all
  // of them resident at once (one round, no tail).  A range that crosses a tile boundary is worked
  // off as two pieces (codebooks restaged, sums flushed in between). ----
  const unsigned total = (unsigned)n_col_tiles * (unsigned)units_total;
  unsigned gpos = (unsigned)bid * (unsigned)units_per_wg;
  unsigned gend = gpos + (unsigned)units_per_wg;
  if (gend > total) gend = total;
  while (gpos < gend) {
  const int ct = (int)(gpos / (unsigned)units_total);
  const int u_beg = (int)(gpos - (unsigned)ct * (unsigned)units_total);
  int u_end = units_total;
  if ((unsigned)(u_end - u_beg) > gend - gpos) u_end = u_beg + (int)(gend - gpos);
  gpos += (unsigned)(u_end - u_beg);
  const int col0 = ct * kTileN;

  // ---- codebook loads (staged row-wise exactly as in dense_role) ----
  float ev[NE];
  {
    int c = col0 + 4 * i16 + st_j;
    if (c > N - 1) c = N - 1;
    const float* src = lut + (size_t)c * L;
    if constexpr (BITS == 4) {
#pragma unroll
      for (int i = 0; i < EPW / 4; ++i) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(src + st_h * EPW + 4 * i);
        ev[4 * i] = t.x; ev[4 * i + 1] = t.y; ev[4 * i + 2] = t.z; ev[4 * i + 3] = t.w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < RPW; ++i) ev[i] = src[2 * (st_h * RPW + i) + (lane >> 5)];
    }
  }
  // ---- a wave's group g of this piece = units u_beg + 4 (wave + WAVES g) + grp ----
  const int n_groups_wg = (u_end - u_beg + 3) / 4;
  const int n_g = n_groups_wg > wave ? (n_groups_wg - wave + WAVES - 1) / WAVES : 0;
  int cidx = col0 / 4 + i16;
  if (cidx > row_stride - 1) cidx = row_stride - 1;
  const uint32_t lane_bytes = 16u * (uint32_t)cidx;
  auto group_unit = [&](int g, int kq) {  // unit of lane row kq in this wave's group g (may be >= u_end)
    return u_beg + 4 * (wave + WAVES * g) + kq;
  };
  auto load_w = [&](int g, u32x4 (&dw)[R]) {
    int u = group_unit(g, grp);
    if (u > u_end - 1) u = u_end - 1;  // clamped re-read inside the slice; its x pieces are zeroed
    if (u < u_beg) u = u_beg;
    if (SQLLM_MFMA_VAR & 32) u = u_beg + grp;  // measurement: every group re-reads the slice's first rows (cache hits)
    const uint32_t off = (uint32_t)(u * R) * row_bytes + lane_bytes;
#pragma unroll
    for (int r = 0; r < R; ++r) dw[r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(qbase + (off + r * row_bytes)));
  };
  // (the pieces of a unit past the slice are zeroed where they are PARKED, not here: a select right
  // behind the load would make the wave wait for it on the spot)
  auto load_x = [&](int g, int ph, f32x4 (&dx)[XL]) {
    int u = group_unit(g, xkq);
    if (u > u_end - 1) u = u_end - 1;
    if (u < u_beg) u = u_beg;
    if ((SQLLM_MFMA_VAR & 16) && g > 0) return;  // measurement: no vec loads after the first group
#pragma unroll
    for (int j = 0; j < XL; ++j) dx[j] = *reinterpret_cast<const f32x4*>(x + xrow[j] + u * KU + 8 * ph);
  };
  u32x4 wa[R], wb[R];
  f32x4 xa[XL], xb[XL];
  load_w(0, wa);
  load_x(0, 0, xa);
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  // ---- stage the codebooks ----
  if constexpr (BITS == 4) {
    float* dst = lds + ((wave & 1) * 4096 + st_h * EPW * ESTRIDE) / 4 + lane;
#pragma unroll
    for (int i = 0; i < EPW; ++i) dst[i * (ESTRIDE / 4)] = ev[i];
  } else {
    float* dst = lds + (st_j * SUBB) / 4 + (lane >> 5) * (ESTRIDE / 4) + (lane & 31);
#pragma unroll
    for (int i = 0; i < RPW; ++i) dst[2 * (st_h * RPW + i) * (ESTRIDE / 4)] = ev[i];
  }
  f32x4 acc[MB][4];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[mb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();  // codebooks staged (and everybody has left the previous piece's slabs)

  // one phase: 8 k's of each lane row against all MB row blocks; v[j][s] = weight s of column 4c + j
  auto phase = [&](const float (&v)[4][8], const f32x4 (&dx)[XL], int g) {
    const bool live = group_unit(g, xkq) < u_end;
#pragma unroll
    for (int j = 0; j < XL; ++j)
      if (!(SQLLM_MFMA_VAR & 8)) *reinterpret_cast<f32x4*>(xt_w + 8 * j * kXtStride) = live ? dx[j] : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 alo[MB], ahi[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      if constexpr (SQLLM_MFMA_VAR & 8) {  // measurement: A operands straight from the loaded registers, no LDS round trip
        alo[mb] = dx[0];
        ahi[mb] = dx[XL - 1];
      } else {
        alo[mb] = *reinterpret_cast<const f32x4*>(xt_r + 16 * mb * kXtStride);
        ahi[mb] = *reinterpret_cast<const f32x4*>(xt_r + 16 * mb * kXtStride + 4);
      }
    }
    if (SQLLM_MFMA_VAR & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const float a = s2 < 4 ? alo[mb][s2 & 3] : ahi[mb][s2 & 3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#if SQLLM_MFMA_FAKE
          acc[mb][j].x = __builtin_fmaf(a, v[j][s2], acc[mb][j].x);  // measurement build: everything but the matrix pipe
#else
          acc[mb][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v[j][s2], acc[mb][j], 0, 0, 0);
#endif
        }
      }
    if (SQLLM_MFMA_VAR & 1) __builtin_amdgcn_s_setprio(0);
  };
  auto lookups = [&](const u32x4 (&t)[R], auto ph_tag, float (&v)[4][8]) {
    constexpr int PH = decltype(ph_tag)::value;
    if constexpr (SQLLM_MFMA_VAR & 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[j][i] = __builtin_bit_cast(float, t[0].x ^ (uint32_t)(j * 8 + i));
      return;
    }
    if constexpr (BITS == 4) {
      const uint32_t w4[4] = {t[0].x, t[0].y, t[0].z, t[0].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t lo = w4[j] & 0x0F0F0F0Fu, hi = (w4[j] >> 4) & 0x0F0F0F0Fu;
        const int off = (j >> 1) * 4096 + (j & 1) * 128;
        v[j][0] = lds_read_f32(__builtin_amdgcn_perm(lo, lane_off, 0x0C0C0400u) + off);
        v[j][1] = lds_read_f32(__builtin_amdgcn_perm(hi, lane_off, 0x0C0C0400u) + off);
        v[j][2] = lds_read_f32(__builtin_amdgcn_perm(lo, lane_off, 0x0C0C0500u) + off);
        v[j][3] = lds_read_f32(__builtin_amdgcn_perm(hi, lane_off, 0x0C0C0500u) + off);
        v[j][4] = lds_read_f32(__builtin_amdgcn_perm(lo, lane_off, 0x0C0C0600u) + off);
        v[j][5] = lds_read_f32(__builtin_amdgcn_perm(hi, lane_off, 0x0C0C0600u) + off);
        v[j][6] = lds_read_f32(__builtin_amdgcn_perm(lo, lane_off, 0x0C0C0700u) + off);
        v[j][7] = lds_read_f32(__builtin_amdgcn_perm(hi, lane_off, 0x0C0C0700u) + off);
      }
    } else {
      const uint32_t t0[4] = {t[0].x, t[0].y, t[0].z, t[0].w};
      const uint32_t t1[4] = {t[1].x, t[1].y, t[1].z, t[1].w};
      const uint32_t t2[4] = {t[2].x, t[2].y, t[2].z, t[2].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j][0] = lds_read_f32(tb[j] | field3_x128<8 * PH + 0>(t0[j], t1[j], t2[j]));
        v[j][1] = lds_read_f32(tb[j] | field3_x128<8 * PH + 1>(t0[j], t1[j], t2[j]));
        v[j][2] = lds_read_f32(tb[j] | field3_x128<8 * PH + 2>(t0[j], t1[j], t2[j]));
        v[j][3] = lds_read_f32(tb[j] | field3_x128<8 * PH + 3>(t0[j], t1[j], t2[j]));
        v[j][4] = lds_read_f32(tb[j] | field3_x128<8 * PH + 4>(t0[j], t1[j], t2[j]));
        v[j][5] = lds_read_f32(tb[j] | field3_x128<8 * PH + 5>(t0[j], t1[j], t2[j]));
        v[j][6] = lds_read_f32(tb[j] | field3_x128<8 * PH + 6>(t0[j], t1[j], t2[j]));
        v[j][7] = lds_read_f32(tb[j] | field3_x128<8 * PH + 7>(t0[j], t1[j], t2[j]));
      }
    }
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using P2 = std::integral_constant<int, 2>;
  using P3 = std::integral_constant<int, 3>;
  // decode group g out of (w, xcur = its phase-0 vec pieces); later phases' pieces are loaded one
  // phase ahead, the NEXT group's weights and phase-0 pieces (into wn / xn) before the first phase
  auto decode_group = [&](int g, const u32x4 (&w)[R], f32x4 (&xcur)[XL], u32x4 (&wn)[R], f32x4 (&xn)[XL]) {
    load_w(g + 1, wn);
    float v[4][8];
    if constexpr (NPH == 1) {
      load_x(g + 1, 0, xn);
      if (!(SQLLM_MFMA_VAR & 2)) __builtin_amdgcn_sched_barrier(0);
      lookups(w, P0{}, v);
      phase(v, xcur, g);
    } else {
      f32x4 xo[XL];
      load_x(g, 1, xo);
      if (!(SQLLM_MFMA_VAR & 2)) __builtin_amdgcn_sched_barrier(0);
      lookups(w, P0{}, v);
      phase(v, xcur, g);
      load_x(g, 2, xcur);
      if (!(SQLLM_MFMA_VAR & 2)) __builtin_amdgcn_sched_barrier(0);
      lookups(w, P1{}, v);
      phase(v, xo, g);
      load_x(g, 3, xo);
      if (!(SQLLM_MFMA_VAR & 2)) __builtin_amdgcn_sched_barrier(0);
      lookups(w, P2{}, v);
      phase(v, xcur, g);
      load_x(g + 1, 0, xn);
      if (!(SQLLM_MFMA_VAR & 2)) __builtin_amdgcn_sched_barrier(0);
      lookups(w, P3{}, v);
      phase(v, xo, g);
    }
    if (!(SQLLM_MFMA_VAR & 2)) __builtin_amdgcn_sched_barrier(0);
  };
  // two register sets swap roles (a copy would make the compiler wait for the loads in flight);
  // groups past the wave's last one re-read valid memory and multiply by zeroed vec pieces
  for (int g = 0; g < n_g; g += 2) {
    decode_group(g, wa, xa, wb, xb);
    decode_group(g + 1, wb, xb, wa, xa);
  }

  // ---- waves meet in LDS, one row block at a time: every wave parks its 16 x 64 partial sums in its
  // slab (plain 16-byte stores: LDS float atomics execute lane by lane -- 16 of them per wave cost
  // 50 us per launch here), the workgroup sums the slabs and issues one atomic per (row, column) ----
  __syncthreads();  // everybody is done with the x tiles
  float* slab = slabs + wave * (16 * 64) + (4 * grp) * 64 + 4 * i16;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    if (mb) __syncthreads();
    *reinterpret_cast<f32x4*>(slab + 0 * 64) = f32x4{acc[mb][0].x, acc[mb][1].x, acc[mb][2].x, acc[mb][3].x};
    *reinterpret_cast<f32x4*>(slab + 1 * 64) = f32x4{acc[mb][0].y, acc[mb][1].y, acc[mb][2].y, acc[mb][3].y};
    *reinterpret_cast<f32x4*>(slab + 2 * 64) = f32x4{acc[mb][0].z, acc[mb][1].z, acc[mb][2].z, acc[mb][3].z};
    *reinterpret_cast<f32x4*>(slab + 3 * 64) = f32x4{acc[mb][0].w, acc[mb][1].w, acc[mb][2].w, acc[mb][3].w};
    __syncthreads();
#pragma unroll
    for (int e = tid; e < 16 * 64; e += WAVES * 64) {
      const int r = m0 + 16 * mb + (e >> 6);
      const int col = col0 + (e & 63);
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) sum += slabs[w * (16 * 64) + e];
      if (r < batch && col < N) acc_add(y + (size_t)r * N + col, sum);
    }
  }
  }  // pieces
}

// (4-bit, <= 32 rows: two workgroups per CU = 4 waves per SIMD -- the second argument keeps the 32-row kernel at 128 VGPRs)
template <int BITS, int MB, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, (BITS == 4 && MB <= 2 && !(SQLLM_MFMA_VAR & 256)) ? 4 : 1)
sqllm_fused_batched(const float* x, const GroupArgs ga) {
  __shared__ __attribute__((aligned(16))) float lds[mfma_lds_floats(BITS, MB, WAVES)];
  const Segment sg = ga.seg[0];  // the whole descriptor in one round of scalar loads (see sqllm_fused_matvec)
  asm volatile("" ::SQLLM_SEG_OPERANDS(sg), "s"(x));
  __builtin_amdgcn_sched_barrier(0);
  const KernelGeom& gm = sg.gm;
  const int m0 = blockIdx.y * 16 * MB;
  dense_role_mfma<BITS, MB, WAVES>(x, reinterpret_cast<const u32x4*>(sg.q), sg.y, sg.lut, gm.K, gm.N, gm.batch, m0,
                                   (int)blockIdx.x, gm.col_tiles, gm.units_total, gm.units_per_wg, lds);
}

// The sparse terms of a wide-batch op, as a launch of their own: inside the matrix-core kernel the
// CSR workgroups would inherit its register allocation (one or two workgroups per CU) and run their
// latency-bound loops without anybody to hide behind -- measured 3.3 ms of a 5.7 ms launch at
// 2048 rows.  Passes of 64 rows (blockIdx.y); blockIdx.x = CSR chunks, then top-X slabs.
//   xT != null: the CSR role reads the transposed copy of vec (lane = batch row);
//   xT == null (no scratch, or the stream is capturing): it gathers from vec, 32 rows at a time.
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
sqllm_sparse_batched(const float* x, const GroupArgs ga, const float* xT, int Bp) {
  constexpr int T = WAVES * 64;
  __shared__ __attribute__((aligned(16))) float lds[cmax(kCsrSpanMax + cmax(kCsrSpanMax, 64 * (kCsrXtSpan + 1)), kTopxLds)];
  const Segment sg = ga.seg[0];  // the whole descriptor in one round of scalar loads (see sqllm_fused_matvec)
  asm volatile("" ::SQLLM_SEG_OPERANDS(sg), "s"(x));
  __builtin_amdgcn_sched_barrier(0);
  const KernelGeom& gm = sg.gm;
  const int sp = blockIdx.x;
  const int m0 = blockIdx.y * 64;
  int rows_here = gm.batch - m0;
  if (rows_here > 64) rows_here = 64;
  if (sp < gm.csr_blocks) {
    if (xT) {
      csr_role<T, 1, float, float, true>(x, sg.y, sg.rows, sg.cols, sg.vals, gm.nnz, gm.K, gm.N, m0, rows_here, sp, lds, nullptr, 0, xT, Bp);
    } else {
      constexpr int CBT = 32;  // every group of rows costs the chunk a zero / gather / flush round with its barriers
      for (int bb = 0; bb < rows_here; bb += CBT) {
        if (bb) __syncthreads();
        csr_role<T, CBT, float, float>(x, sg.y, sg.rows, sg.cols, sg.vals, gm.nnz, gm.K, gm.N, m0 + bb,
                                       rows_here - bb < CBT ? rows_here - bb : CBT, sp, lds, nullptr, 0);
      }
    }
  } else if (sp < gm.csr_blocks + gm.topx_blocks) {
    topx_role<T, float, float>(x, sg.y, sg.full_rows, sg.full_idx, gm.topX, gm.K, gm.N, m0, rows_here, sp - gm.csr_blocks, lds);
  }
}

// ------------------------------------------------------------------------------------------------
// Small-batch dense role (the *_batched operators from 2 rows up to the matrix-core switch-over):
// lane = OUTPUT COLUMN, every k wave-uniform.
//
// The batch tiles of dense_role broadcast vec across 16-lane rows (one DPP move + half a packed FMA
// per weight and batch row) and the matrix-core kernel cannot hide its decode behind the fp32
// matrix instructions (on gfx950 v_mfma_f32_16x16x4_f32 and vector / LDS work ADD:
// tools/experiments/issue_rate.hip, "phases" rows), so between 2 and ~16 rows both cost 2-3 x the
// batch-1 launch.  Here a lane owns one column:
//     * a wave load is 64 columns x one qweight row (256 contiguous bytes); wave w of the workgroup
//       takes units u0 + w, u0 + w + WAVES, ... of the piece, D units per load batch, two batches in
//       flight (raw buffer loads whose descriptor ends with the piece: a unit past it returns 0
//       without touching memory, so the loop has one static shape);
//     * vec is WAVE-UNIFORM: it comes from SGPRs (s_load through the constant address space) and
//       is the scalar operand of v_pk_fma_f32 -- no DPP, no LDS traffic, no VALU instruction for
//       vec at all; per weight and batch row the loop issues HALF a vector instruction;
//     * 4-bit: 16-entry codebooks transposed in LDS ([entry][column], 4 KB: any 64 lookups are
//       conflict-free), address = one v_perm_b32 per weight; 3-bit: 64-entry PAIR tables
//       ([i0 + 8 i1][column] x {lut[i0], lut[i1]}, 32 KB): one ds_read_b64 per two weights;
//     * decode is a software pipeline over stages of ST k's (ST * rows <= 32 SGPRs of vec):
//       [wait] [issue lookups + vec loads of stage s + 1] [packed FMAs of stage s]; scalar loads
//       return out of order, so the wait is lgkmcnt(0) and sits in front of the next issue.
// Work is cut as in the matrix-core kernel: equal contiguous ranges of the flattened
// (column tile, unit) space, one per workgroup, a range crossing a tile boundary = two pieces.
// Waves meet in LDS slabs (plain stores) and leave as one atomic per (row, column).
// ------------------------------------------------------------------------------------------------
typedef float f32x8 __attribute__((ext_vector_type(8)));
// one extra, all-zero entry row behind each table: a unit past the end of a wave's share looks it up
constexpr int cols_table_floats(int bits) { return bits == 4 ? 17 * 64 : 65 * 64 * 2; }
constexpr int cols_lds_floats(int bits, int bt, int waves) {
  return cmax(cols_table_floats(bits) + waves * bt * 64, cmax(2 * kCsrSpanMax, kTopxLds));
}
constexpr int cols_stage_k(int bt) { return bt * 8 <= 32 ? 8 : 4; }  // k's per pipeline stage

// 6-bit field M (weights 2M, 2M + 1 of a 3-bit unit) times 512, ready to be OR-ed into a pair-table address
__device__ __forceinline__ uint32_t field6_x512(uint32_t t0, uint32_t t1, uint32_t t2, int M) {
  const int bit = 6 * M, w = bit >> 5, o = bit & 31;
  const uint32_t lo = (w == 0) ? t0 : (w == 1) ? t1 : t2;
  uint32_t f;
  if (o <= 26) {
    if (o > 9) f = lo >> (o - 9);
    else if (o < 9) f = lo << (9 - o);
    else f = lo;
  } else {
    const uint32_t hi = (w == 0) ? t1 : t2;
    f = __builtin_amdgcn_alignbit(hi, lo, o) << 9;
  }
  return f & 0x7E00u;
}

// vec -> SGPRs: scalar loads through the constant address space, base (one SGPR pair) + byte offset.
typedef const __attribute__((address_space(4))) float* cfloatp;
typedef const __attribute__((address_space(4))) char* ccharp;
template <int ST> struct XVec;
template <> struct XVec<4> {
  typedef f32x4 type;
  static __device__ __forceinline__ type load(ccharp base, uint32_t off) {
    return *reinterpret_cast<const __attribute__((address_space(4))) type*>(base + off);  // (16-byte aligned: K % 32 == 0)
  }
};
template <> struct XVec<8> {
  typedef f32x8 type;
  static __device__ __forceinline__ type load(ccharp base, uint32_t off) {
    return *reinterpret_cast<const __attribute__((address_space(4))) type*>(base + off);
  }
};

template <int BITS, int BT, int WAVES>
__device__ __forceinline__ void dense_role_cols(const float* __restrict__ x, const uint32_t* __restrict__ q,
                                                float* __restrict__ y, const float* __restrict__ lut, int K, int N,
                                                int b0, int nb, int bid, int n_col_tiles, int units_total,
                                                int units_per_wg, float* lds) {
  using F = Fmt<BITS>;
  constexpr int L = F::kLut, R = F::kRows, KU = F::kK;
  constexpr int ST = cols_stage_k(BT);        // k's per stage
  constexpr int SPU = KU / ST;                // stages per unit
  constexpr int D = BITS == 4 ? 4 : 2;        // units per load batch ("chunk")
  constexpr int NB = 2;                       // chunks in flight
  constexpr int NS = D * SPU;                 // stages per chunk
  constexpr int NP = ST / 2;                  // weight pairs per stage
  constexpr int NA = BT >= 4 ? 1 : 4 / BT;    // accumulators per batch row (FMA chains in flight)
  using XV = typename XVec<ST>::type;
  static_assert(NS % 2 == 0, "stages alternate between two register sets");
  static_assert(WAVES == 8, "table build assumes 8 waves");
  __builtin_amdgcn_s_waitcnt(0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* table = reinterpret_cast<char*>(lds);
  float* slabs = lds + cols_table_floats(BITS);
  // 4-bit: byte 1 of the lane word is the index of the zero row -- a lookup's v_perm takes its entry
  // index from the data (selector 4..7) or, for a unit that does not exist, from here (selector 1)
  const uint32_t lane_base = BITS == 4 ? (4u * lane) | 0x1000u : 8u * lane;
  const uint32_t row_bytes = 4u * (uint32_t)N;
  // rows of this pass (rows past the batch re-read its last row; never stored): one base pointer each,
  // so that a vec load is  base + the stage's byte offset  with no scalar arithmetic of its own
  ccharp xrow[BT];
#pragma unroll
  for (int b = 0; b < BT; ++b) xrow[b] = reinterpret_cast<ccharp>(reinterpret_cast<uintptr_t>(x + (size_t)(b0 + (b < nb ? b : nb - 1)) * K));
  uint32_t soff[D][R];
#pragma unroll
  for (int j = 0; j < D; ++j)
#pragma unroll
    for (int r = 0; r < R; ++r) soff[j][r] = (uint32_t)(j * WAVES * R + r) * row_bytes;
  // the zero rows of the tables (never rewritten)
  if constexpr (BITS == 4) { if (tid < 64) *reinterpret_cast<float*>(table + 16 * 256 + 4 * tid) = 0.f; }
  else { if (tid < 64) *reinterpret_cast<f32x2*>(table + 64 * 512 + 8 * tid) = f32x2{0.f, 0.f}; }

  const unsigned total = (unsigned)n_col_tiles * (unsigned)units_total;
  unsigned gpos = (unsigned)bid * (unsigned)units_per_wg;
  unsigned gend = gpos + (unsigned)units_per_wg;
  if (gend > total) gend = total;
  bool first = true;
  while (gpos < gend) {
    const int ct = __builtin_amdgcn_readfirstlane((int)(gpos / (unsigned)units_total));  // (the division runs on the VALU)
    const int u0 = (int)(gpos - (unsigned)ct * (unsigned)units_total);
    int u1 = units_total;
    if ((unsigned)(u1 - u0) > gend - gpos) u1 = u0 + (int)(gend - gpos);
    gpos += (unsigned)(u1 - u0);
    const int col0 = ct * kTileN;
    const int col = col0 + lane;
    const int colc = col < N ? col : N - 1;
    // ---- codebook values for the table rows this wave builds, then the first NB chunks ----
    const float* lp = lut + (size_t)colc * L;
    float tv[8];
    float thi = 0.f;
    if constexpr (BITS == 4) {
      const f32x2 t = *reinterpret_cast<const f32x2*>(lp + 2 * w);
      tv[0] = t.x; tv[1] = t.y;
    } else {
      const f32x4 ta = *reinterpret_cast<const f32x4*>(lp), tb2 = *reinterpret_cast<const f32x4*>(lp + 4);
      tv[0] = ta.x; tv[1] = ta.y; tv[2] = ta.z; tv[3] = ta.w; tv[4] = tb2.x; tv[5] = tb2.y; tv[6] = tb2.z; tv[7] = tb2.w;
      thi = lp[w];
    }
    const int n_units = u1 - u0;
    const int n_w = n_units > w ? (n_units - w + WAVES - 1) / WAVES : 0;  // units of this wave: u0 + w + WAVES i
    const int nc = (n_w + D - 1) / D;
    const __amdgpu_buffer_rsrc_t qrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(q), 0, (uint32_t)(u1 * R) * row_bytes, 0x00020000);
    uint32_t voff[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) voff[k] = 4u * (uint32_t)colc + (uint32_t)((u0 + w + k * D * WAVES) * R) * row_bytes;
    auto load_chunk = [&](int k, uint32_t (&dst)[D][R]) {
#pragma unroll
      for (int j = 0; j < D; ++j)
#pragma unroll
        for (int r = 0; r < R; ++r) dst[j][r] = __builtin_amdgcn_raw_buffer_load_b32(qrsrc, voff[k], soff[j][r], 2 /* nt */);
      voff[k] += (uint32_t)(NB * D * WAVES * R) * row_bytes;
    };
    uint32_t wbuf[NB][D][R];
#pragma unroll
    for (int k = 0; k < NB; ++k) load_chunk(k, wbuf[k]);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (!first) __syncthreads();  // everybody has left the previous piece's table and slabs
    first = false;
    if constexpr (BITS == 4) {
      *reinterpret_cast<float*>(table + (2 * w) * 256 + 4 * lane) = tv[0];
      *reinterpret_cast<float*>(table + (2 * w + 1) * 256 + 4 * lane) = tv[1];
    } else {
#pragma unroll
      for (int i0 = 0; i0 < 8; ++i0) *reinterpret_cast<f32x2*>(table + (8 * w + i0) * 512 + lane_base) = f32x2{tv[i0], thi};
    }
    __syncthreads();

    f32x2 acc[BT][NA];
#pragma unroll
    for (int b = 0; b < BT; ++b)
#pragma unroll
      for (int a = 0; a < NA; ++a) acc[b][a] = f32x2{0.f, 0.f};
    struct St { f32x2 v[NP]; XV x[BT]; };
    // issue stage st of chunk cc (held in buf): lookups + vec loads.  A unit past the wave's last one
    // (its weight words are zeros, but entry 0 of a codebook need not be) looks up the zero row -- the
    // choice is a scalar select of the v_perm selector (4-bit) or one OR per unit (3-bit) -- and
    // re-reads the vec of the piece's last unit.
    auto issue = [&](const uint32_t (&buf)[D][R], int st, int cc, St& o) {
      const int j = st / SPU, sub = st % SPU;
      const int i = cc * D + j;
      const bool live = i < n_w;
      if constexpr (BITS == 4) {
        const uint32_t lo = buf[j][0] & 0x0F0F0F0Fu, hi = (buf[j][0] >> 4) & 0x0F0F0F0Fu;
#define SQ_L(WORD, SEL) *reinterpret_cast<const float __attribute__((address_space(3)))*>(__builtin_amdgcn_perm(WORD, lane_base, SEL))
#pragma unroll
        for (int m = 0; m < NP; ++m) {
          const uint32_t sel = live ? 0x0C0C0400u + ((uint32_t)(sub * NP + m) << 8) : 0x0C0C0100u;
          o.v[m] = f32x2{SQ_L(lo, sel), SQ_L(hi, sel)};
        }
#undef SQ_L
      } else {
        const uint32_t lane_or_dead = lane_base | (live ? 0u : 0x8000u);
#pragma unroll
        for (int m = 0; m < NP; ++m)
          o.v[m] = *reinterpret_cast<const f32x2 __attribute__((address_space(3)))*>(
              lane_or_dead | field6_x512(buf[j][0], buf[j][1], buf[j][2], sub * NP + m));
      }
      int u = u0 + w + i * WAVES;
      if (u > u1 - 1) u = u1 - 1;
      uint32_t koff = 4u * (uint32_t)(u * KU + sub * ST);
      asm volatile("" : "+s"(koff));  // (keeps the compiler from turning the 8 offsets into 8 induction variables)
#pragma unroll
      for (int b = 0; b < BT; ++b) o.x[b] = XVec<ST>::load(xrow[b], koff);
    };
    auto fmas = [&](const St& o) {
#pragma unroll
      for (int m = 0; m < NP; ++m)
#pragma unroll
        for (int b = 0; b < BT; ++b)
          acc[b][m % NA] = __builtin_elementwise_fma(o.v[m], f32x2{o.x[b][2 * m], o.x[b][2 * m + 1]}, acc[b][m % NA]);
    };
    St sa, sb;
    issue(wbuf[0], 0, 0, sa);
    __builtin_amdgcn_sched_barrier(0);
    for (int c = 0; c < nc; c += NB) {
#pragma unroll
      for (int k = 0; k < NB; ++k) {
#pragma unroll
        for (int st = 0; st < NS; st += 2) {
          // the wait for stage s goes BEFORE the issue of stage s + 1 (behind it, it would wait for the new lookups too)
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

    // ---- waves meet in LDS: slab [wave][row][column], then one atomic per (row, column) ----
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      f32x2 t = acc[b][0];
#pragma unroll
      for (int a = 1; a < NA; ++a) t += acc[b][a];
      slabs[(w * BT + b) * 64 + lane] = t.x + t.y;
    }
    __syncthreads();
    for (int e = tid; e < BT * 64; e += WAVES * 64) {
      const int b = e >> 6, c = e & 63;
      float sum = 0.f;
#pragma unroll
      for (int ww = 0; ww < WAVES; ++ww) sum += slabs[(ww * BT + b) * 64 + c];
      if (b < nb && col0 + c < N) acc_add(y + (size_t)(b0 + b) * N + col0 + c, sum);
    }
  }
}

template <int BITS, int BT, int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
sqllm_fused_cols(const float* x, const GroupArgs ga) {
  constexpr int T = WAVES * 64;
  __shared__ __attribute__((aligned(16))) float lds[cols_lds_floats(BITS, BT, WAVES)];
  const Segment sg = ga.seg[0];  // the whole descriptor in one round of scalar loads (see sqllm_fused_matvec)
  asm volatile("" ::SQLLM_SEG_OPERANDS(sg), "s"(x));
  __builtin_amdgcn_sched_barrier(0);
  const KernelGeom& gm = sg.gm;
  const int bid = blockIdx.x;
  const int b0 = blockIdx.y * BT;
  int nb = gm.batch - b0;
  if (nb > BT) nb = BT;
  const int d = bid - gm.dense_block0;
  const int sp = bid < gm.dense_block0 ? bid : -1;
  if (d >= 0 && d < gm.dense_blocks) {
    dense_role_cols<BITS, BT, WAVES>(x, sg.q, sg.y, sg.lut, gm.K, gm.N, b0, nb, d, gm.col_tiles, gm.units_total,
                                     gm.units_per_wg, lds);
  } else if (sp >= 0 && sp < gm.csr_blocks) {
    csr_role<T, BT, float, float>(x, sg.y, sg.rows, sg.cols, sg.vals, gm.nnz, gm.K, gm.N, b0, nb, sp, lds, nullptr, 0);
  } else if (sp >= gm.csr_blocks && sp < gm.csr_blocks + gm.topx_blocks) {
    topx_role<T, float, float>(x, sg.y, sg.full_rows, sg.full_idx, gm.topX, gm.K, gm.N, b0, nb, sp - gm.csr_blocks, lds);
  }
}

#ifdef SQLLM_ABLATION_BUILD
// calibration kernels (measurement builds only): what does this box give an empty launch and a
// plain linear 16-B/lane streaming read of the same bytes?
__global__ void __launch_bounds__(256) sqllm_calib_empty(float* y) {
  if (threadIdx.x == 12345) y[0] = 1.f;
}
template <int UNROLL, bool NT>
__global__ void __launch_bounds__(256) sqllm_calib_stream(const u32x4* q, size_t n16, float* y) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t acc = 0;
  for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
    u32x4 w[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) w[u] = NT ? __builtin_nontemporal_load(q + i + u * stride) : q[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= w[u].x ^ w[u].y ^ w[u].z ^ w[u].w;
  }
  for (; i < n16; i += stride) { u32x4 w = q[i]; acc ^= w.x ^ w.y ^ w.z ^ w.w; }
  if (acc == 0x12345678u) y[0] = 1.f;
}
// tiled streaming read: a wave covers (64 / SEGL) rows x (SEGL lanes x 16 B) per load instruction,
// a workgroup of 4 waves walks `rows_per_wg` rows of one column tile -- how narrow may a row segment
// get before HBM efficiency drops?
template <int SEGL>
__global__ void __launch_bounds__(256) sqllm_calib_tiled(const u32x4* q, int rows_total, int row_stride16,
                                                        int col_tiles, int rows_per_wg, float* y) {
  constexpr int RPI = 64 / SEGL;  // rows per wave-instruction
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ct = blockIdx.x % col_tiles, ks = blockIdx.x / col_tiles;
  int c16 = ct * SEGL + (lane % SEGL);
  if (c16 > row_stride16 - 1) c16 = row_stride16 - 1;
  const int r0 = ks * rows_per_wg;
  int r1 = r0 + rows_per_wg;
  if (r1 > rows_total) r1 = rows_total;
  uint32_t acc = 0;
  // wave w takes rows r0 + w*RPI + lane/SEGL, stepping 4*RPI
  for (int r = r0 + wave * RPI + lane / SEGL; r < r1; r += 4 * RPI * 4) {
    u32x4 w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int rr = r + u * 4 * RPI;
      if (rr > rows_total - 1) rr = rows_total - 1;
      w[u] = __builtin_nontemporal_load(q + (size_t)rr * row_stride16 + c16);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc ^= w[u].x ^ w[u].y ^ w[u].z ^ w[u].w;
  }
  if (acc == 0x12345678u) y[0] = 1.f;
}
template <int SEGL>
static void launch_tiled(const LaunchArgs& a, hipStream_t stream, int target_wgs) {
  const int rows_total = a.ga.seg[0].gm.units_total * (a.ga.seg[0].gm.K / a.ga.seg[0].gm.units_total == 8 ? 1 : 3);
  const int row_stride16 = a.ga.seg[0].gm.N / 4;
  const int col_tiles = (row_stride16 + SEGL - 1) / SEGL;
  int slices = (target_wgs + col_tiles - 1) / col_tiles;
  if (slices < 1) slices = 1;
  int rows_per_wg = (rows_total + slices - 1) / slices;
  const int gran = 16 * (64 / SEGL);
  rows_per_wg = (rows_per_wg + gran - 1) / gran * gran;
  slices = (rows_total + rows_per_wg - 1) / rows_per_wg;
  hipExtLaunchKernelGGL((sqllm_calib_tiled<SEGL>), dim3(col_tiles * slices), dim3(256), 0, stream, a.ev_start, a.ev_stop, 0,
                        reinterpret_cast<const u32x4*>(a.ga.seg[0].q), rows_total, row_stride16, col_tiles, rows_per_wg, a.ga.seg[0].y);
}
static hipError_t launch_calib(const LaunchArgs& a, hipStream_t stream) {
  if (a.ablate >= 200) {  // 2SW: S = log2(lanes per segment) - 3 (0..3 -> 8,16,32,64 lanes), W = target wgs / 256
    const int sg = (a.ablate / 10) % 10, tw = (a.ablate % 10) * 256;
    if (sg == 0) launch_tiled<8>(a, stream, tw);
    else if (sg == 1) launch_tiled<16>(a, stream, tw);
    else if (sg == 2) launch_tiled<32>(a, stream, tw);
    else launch_tiled<64>(a, stream, tw);
    return hipGetLastError();
  }
  const size_t n16 = (size_t)a.ga.seg[0].gm.units_total * (a.ga.seg[0].gm.K / a.ga.seg[0].gm.units_total == 8 ? 1 : 3) * (a.ga.seg[0].gm.N / 4);
  const int mode = a.ablate;
  dim3 grid(mode == 100 ? 512 : (mode % 10 == 1 ? 512 : mode % 10 == 2 ? 1024 : mode % 10 == 3 ? 2048 : 4096));
  if (mode == 100) hipExtLaunchKernelGGL(sqllm_calib_empty, grid, dim3(256), 0, stream, a.ev_start, a.ev_stop, 0, a.ga.seg[0].y);
  else if (mode < 120) hipExtLaunchKernelGGL((sqllm_calib_stream<4, true>), grid, dim3(256), 0, stream, a.ev_start, a.ev_stop, 0, reinterpret_cast<const u32x4*>(a.ga.seg[0].q), n16, a.ga.seg[0].y);
  else if (mode < 130) hipExtLaunchKernelGGL((sqllm_calib_stream<8, true>), grid, dim3(256), 0, stream, a.ev_start, a.ev_stop, 0, reinterpret_cast<const u32x4*>(a.ga.seg[0].q), n16, a.ga.seg[0].y);
  else hipExtLaunchKernelGGL((sqllm_calib_stream<8, false>), grid, dim3(256), 0, stream, a.ev_start, a.ev_stop, 0, reinterpret_cast<const u32x4*>(a.ga.seg[0].q), n16, a.ga.seg[0].y);
  return hipGetLastError();
}
#endif

template <int BITS, int BT, int WAVES, int ABL = 0, bool LIN = false>
static hipError_t launch_inst(const LaunchArgs& a, hipStream_t stream) {
  const int batch = a.ga.seg[0].gm.batch;
  dim3 grid(a.ga.block0[a.ga.n_seg], (batch + BT - 1) / BT);
  auto kern = sqllm_fused_matvec<BITS, BT, WAVES, ABL, LIN>;
  if (a.ev_start || a.ev_stop) {
    // same kernel, with the dispatch's own begin/end timestamps exposed through two events
    hipExtLaunchKernelGGL(kern, grid, dim3(WAVES * 64), 0, stream, a.ev_start, a.ev_stop, 0, a.x, a.ga);
  } else {
    hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), 0, stream, a.x, a.ga);
  }
  return hipGetLastError();
}

template <int BITS, bool LIN>
static hipError_t launch_bt(const LaunchArgs& a, hipStream_t stream) {
  switch (batch_tile(a.ga.seg[0].gm.batch)) {
    case 1: return launch_inst<BITS, 1, kWaves, 0, LIN>(a, stream);
    case 2: return launch_inst<BITS, 2, kWaves, 0, LIN>(a, stream);
    case 4: return launch_inst<BITS, 4, kWaves, 0, LIN>(a, stream);
    default: return launch_inst<BITS, 8, kWaves, 0, LIN>(a, stream);
  }
}

template <int BITS, int MB>
static hipError_t launch_mfma_inst(const LaunchArgs& a, hipStream_t stream) {
  const KernelGeom& gm = a.ga.seg[0].gm;
  dim3 grid(gm.dense_blocks, (gm.batch + 16 * MB - 1) / (16 * MB));
  auto kern = sqllm_fused_batched<BITS, MB, kWaves>;
  const float* x = static_cast<const float*>(a.x);
  if (a.ev_start || a.ev_stop) hipExtLaunchKernelGGL(kern, grid, dim3(kWaves * 64), 0, stream, a.ev_start, a.ev_stop, 0, x, a.ga);
  else hipLaunchKernelGGL(kern, grid, dim3(kWaves * 64), 0, stream, x, a.ga);
  return hipGetLastError();
}

template <int BITS>
static hipError_t launch_mfma_bits(const LaunchArgs& a, hipStream_t stream) {
  switch (mfma_row_blocks(a.ga.seg[0].gm.batch)) {
    case 1: return launch_mfma_inst<BITS, 1>(a, stream);
    case 2: return launch_mfma_inst<BITS, 2>(a, stream);
    default: return launch_mfma_inst<BITS, 4>(a, stream);
  }
}

template <int BITS, int BT>
static hipError_t launch_cols_inst(const LaunchArgs& a, hipStream_t stream) {
  const KernelGeom& gm = a.ga.seg[0].gm;
  dim3 grid(gm.dense_block0 + gm.dense_blocks, (gm.batch + BT - 1) / BT);
  auto kern = sqllm_fused_cols<BITS, BT, kWaves>;
  const float* x = static_cast<const float*>(a.x);
  if (a.ev_start || a.ev_stop) hipExtLaunchKernelGGL(kern, grid, dim3(kWaves * 64), 0, stream, a.ev_start, a.ev_stop, 0, x, a.ga);
  else hipLaunchKernelGGL(kern, grid, dim3(kWaves * 64), 0, stream, x, a.ga);
  return hipGetLastError();
}

template <int BITS>
static hipError_t launch_cols_bits(const LaunchArgs& a, hipStream_t stream) {
  switch (batch_tile(a.ga.seg[0].gm.batch)) {
    case 1: return launch_cols_inst<BITS, 1>(a, stream);
    case 2: return launch_cols_inst<BITS, 2>(a, stream);
    case 4: return launch_cols_inst<BITS, 4>(a, stream);
    default: return launch_cols_inst<BITS, 8>(a, stream);
  }
}

// one op (a.ga.seg[0]), operator ABI, small batches: lane = column, vec from SGPRs
hipError_t launch_batched_cols(int bits, const LaunchArgs& a, hipStream_t stream) {
  return bits == 4 ? launch_cols_bits<4>(a, stream) : launch_cols_bits<3>(a, stream);
}

// the CSR and top-X terms of one wide-batch op (a.ga.seg[0]); a.xT = transposed vec or null
hipError_t launch_batched_sparse(const LaunchArgs& a, hipStream_t stream) {
  const KernelGeom& gm = a.ga.seg[0].gm;
  if (gm.csr_blocks + gm.topx_blocks <= 0) return hipSuccess;
  dim3 grid(gm.csr_blocks + gm.topx_blocks, (gm.batch + 63) / 64);
  auto kern = sqllm_sparse_batched<kWaves>;
  const float* x = static_cast<const float*>(a.x);
  if (a.ev_start || a.ev_stop) hipExtLaunchKernelGGL(kern, grid, dim3(kWaves * 64), 0, stream, a.ev_start, a.ev_stop, 0, x, a.ga, a.xT, a.Bp);
  else hipLaunchKernelGGL(kern, grid, dim3(kWaves * 64), 0, stream, x, a.ga, a.xT, a.Bp);
  return hipGetLastError();
}

// one op (a.ga.seg[0]), operator ABI, batch rows through the matrix cores (dense term only)
hipError_t launch_batched_mfma(int bits, const LaunchArgs& a, hipStream_t stream) {
  return bits == 4 ? launch_mfma_bits<4>(a, stream) : launch_mfma_bits<3>(a, stream);
}

// vec [batch, K] -> xT [K, Bp] (Bp = batch rounded up to 64; the padding rows are zeros) for the
// wide-batch CSR role: 64 x 64 tiles through LDS, reads coalesced along k, writes along the rows.
__global__ void __launch_bounds__(256) sqllm_transpose_vec(const float* __restrict__ x, float* __restrict__ xT, int batch, int K, int Bp) {
  __shared__ float tile[64][65];
  const int k0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = r0 + ty + 4 * i, k = k0 + tx;
    tile[ty + 4 * i][tx] = (r < batch && k < K) ? x[(size_t)r * K + k] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int k = k0 + ty + 4 * i, r = r0 + tx;
    if (k < K) xT[(size_t)k * Bp + r] = tile[tx][ty + 4 * i];
  }
}

hipError_t transpose_vec(const float* x, float* xT, int batch, int K, int Bp, hipStream_t stream, hipEvent_t ev_start) {
  if (ev_start) hipExtLaunchKernelGGL(sqllm_transpose_vec, dim3((K + 63) / 64, Bp / 64), dim3(256), 0, stream, ev_start, nullptr, 0, x, xT, batch, K, Bp);
  else hipLaunchKernelGGL(sqllm_transpose_vec, dim3((K + 63) / 64, Bp / 64), dim3(256), 0, stream, x, xT, batch, K, Bp);
  return hipGetLastError();
}

// Debug aid (option "validate_csr"): is `rows` a CSR row-pointer array for nnz values?  The fused
// linear detects completion by counting the contributions `rows` announces, so an inconsistent
// array leaves columns unfinished and the workspace dirty; this check makes that a loud error.
// Blocks the host (one tiny kernel + a 4-byte read-back); skipped while the stream is capturing.
__global__ void sqllm_check_csr(const int* __restrict__ rows, int N, int nnz, int* flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && (rows[0] != 0 || rows[N] != nnz)) atomicOr(flag, 1);
  if (i < N && rows[i + 1] < rows[i]) atomicOr(flag, 2);
}

hipError_t check_csr(const int* rows, int N, int nnz, hipStream_t stream, int* bad) {
  *bad = 0;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return hipSuccess;
  // The 4-byte flag is stream-ordered scratch of the CURRENT device (the option is per device and
  // several GPUs can be driven from one process: a process-wide buffer would live on whichever device
  // used the option first).  Debug path: the allocation cost does not matter.
  int* flag = nullptr;
  hipError_t e;
  if ((e = hipMallocAsync(reinterpret_cast<void**>(&flag), sizeof(int), stream)) != hipSuccess) return e;
  if ((e = hipMemsetAsync(flag, 0, sizeof(int), stream)) == hipSuccess) {
    hipLaunchKernelGGL(sqllm_check_csr, dim3((N + 255) / 256), dim3(256), 0, stream, rows, N, nnz, flag);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(bad, flag, sizeof(int), hipMemcpyDeviceToHost, stream);
  (void)hipFreeAsync(flag, stream);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(stream);
}

hipError_t launch_fused(int bits, const LaunchArgs& a, hipStream_t stream) {
#ifdef SQLLM_ABLATION_BUILD
  if (a.ablate >= 100) return launch_calib(a, stream);
  if (!a.linear && bits == 4 && batch_tile(a.ga.seg[0].gm.batch) == 1 && a.ablate) {
    switch (a.ablate) {
      case 1: return launch_inst<4, 1, kWaves, 1>(a, stream);
      case 2: return launch_inst<4, 1, kWaves, 2>(a, stream);
      case 4: return launch_inst<4, 1, kWaves, 4>(a, stream);
      case 8: return launch_inst<4, 1, kWaves, 8>(a, stream);
      case 13: return launch_inst<4, 1, kWaves, 13>(a, stream);
      case 14: return launch_inst<4, 1, kWaves, 14>(a, stream);
      case 16: return launch_inst<4, 1, kWaves, 16>(a, stream);
      case 32: return launch_inst<4, 1, kWaves, 32>(a, stream);
      case 40: return launch_inst<4, 1, kWaves, 128>(a, stream);  // option value 40 = ABL bit 128
      default: break;
    }
  }
#endif
  if (a.linear) return bits == 4 ? launch_bt<4, true>(a, stream) : launch_bt<3, true>(a, stream);
  return bits == 4 ? launch_bt<4, false>(a, stream) : launch_bt<3, false>(a, stream);
}

}  // namespace sqllm
