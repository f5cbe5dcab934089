// sqllm_split_common.h -- what the split matrix-core kernels share (sqllm_mfma_split.hip: the tile form and the fused
// small launch; sqllm_mfma_wide.hip: the wide form): the exact three-way bf16 split of fp32 values, the split codebook
// entries in LDS and their lookups, one phase of matrix instructions on split operands.  The arithmetic is described at
// the head of sqllm_mfma_split.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sqllm_kernels.h"

#include "sqllm_decode.h"

namespace sqllm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int split_codebook_bytes(int bits) { return 4 * (1 << bits) * 256; }  // [4 columns][index][32 slots x 8 B]
constexpr int split_lds_floats(int bits, int waves) {
  // codebooks, then the epilogue's slabs [waves][16][64]
  return split_codebook_bytes(bits) / 4 + waves * 16 * 64;
}

__device__ __forceinline__ u32x2 lds_read_u32x2(uint32_t byte_addr) {
  return *reinterpret_cast<const u32x2 __attribute__((address_space(3)))*>(byte_addr);
}

// 3-bit field KIDX of a unit's 96-bit stream, shifted to bit 8 (an entry row is 256 bytes here)
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

// exact three-way split of eight fp32 values into packed bf16 operand registers
__device__ __forceinline__ void split8(const float (&v)[8], uint32_t (&h)[4], uint32_t (&m)[4], uint32_t (&l)[4]) {
  uint32_t hb[8], mb[8], lb[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint32_t b = __builtin_bit_cast(uint32_t, v[i]);
    hb[i] = b & 0xFFFF0000u;
    const float r1 = v[i] - __builtin_bit_cast(float, hb[i]);  // exact
    mb[i] = __builtin_bit_cast(uint32_t, r1) & 0xFFFF0000u;
    const float r2 = r1 - __builtin_bit_cast(float, mb[i]);    // exact, <= 8 significant bits
    lb[i] = __builtin_bit_cast(uint32_t, r2);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // element 2i in the low half, 2i + 1 in the high half
    h[i] = __builtin_amdgcn_perm(hb[2 * i + 1], hb[2 * i], 0x07060302u);
    m[i] = __builtin_amdgcn_perm(mb[2 * i + 1], mb[2 * i], 0x07060302u);
    l[i] = __builtin_amdgcn_perm(lb[2 * i + 1], lb[2 * i], 0x07060302u);
  }
}

__device__ __forceinline__ bf16x8 as_frag(const uint32_t (&r)[4]) {
  typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(bf16x8, u32x4v{r[0], r[1], r[2], r[3]});
}

// One phase of a wave's group: the 8 k's of each lane row (phase PH of its unit) against all MB row blocks.
//   t        the lane's packed words of the unit (4 columns x R rows)
//   dx       the phase's vec values: XMODE 0 fp32 (two registers per row block, split here), 2 / 3 ready-made planes
//   lane_off byte offset of this lane's slot inside an entry row, plus the table's base (3-bit: all of it; 4-bit: bits
//            16.. of it in byte 1 -- the low 16 bits of a 4-bit table's base arrive through `wmask`, OR-ed into the index
//            bytes: bases are multiples of 16 KB, an index is < 16)
template <int BITS, int MB, int XMODE, int PH>
__device__ __forceinline__ void split_phase(const u32x4 (&t)[Fmt<BITS>::kRows], const u32x4 (&dx)[XMODE == 0 ? 2 * MB : XMODE * MB],
                                            bool live, uint32_t lane_off, uint32_t wmask, f32x4 (&acc)[MB][4]) {
  // A fragments of every row block
  uint32_t ah[MB][4], am[MB][4], al[MB][4];
  if constexpr (XMODE == 0) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const f32x4 lo4 = __builtin_bit_cast(f32x4, dx[2 * mb]), hi4 = __builtin_bit_cast(f32x4, dx[2 * mb + 1]);
      float v[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = live ? v[i] : 0.f;
      split8(v, ah[mb], am[mb], al[mb]);
    }
  } else {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const u32x4 h4 = dx[XMODE * mb], m4 = dx[XMODE * mb + 1], l4 = dx[XMODE * mb + XMODE - 1];
      ah[mb][0] = h4.x; ah[mb][1] = h4.y; ah[mb][2] = h4.z; ah[mb][3] = h4.w;
      am[mb][0] = m4.x; am[mb][1] = m4.y; am[mb][2] = m4.z; am[mb][3] = m4.w;
      al[mb][0] = l4.x; al[mb][1] = l4.y; al[mb][2] = l4.z; al[mb][3] = l4.w;  // (XMODE 2: not used)
    }
  }
  uint32_t t0[4], t1[4], t2[4];
  if constexpr (BITS == 4) {
    t0[0] = t[0].x; t0[1] = t[0].y; t0[2] = t[0].z; t0[3] = t[0].w;
  } else {
    t0[0] = t[0].x; t0[1] = t[0].y; t0[2] = t[0].z; t0[3] = t[0].w;
    t1[0] = t[1].x; t1[1] = t[1].y; t1[2] = t[1].z; t1[3] = t[1].w;
    t2[0] = t[2].x; t2[1] = t[2].y; t2[2] = t[2].z; t2[3] = t[2].w;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    // the 8 weights of column 4c + j: one ds_read_b64 each
    u32x2 e[8];
    if constexpr (BITS == 4) {
      const uint32_t lo = (t0[j] & 0x0F0F0F0Fu) | wmask, hi = ((t0[j] >> 4) & 0x0F0F0F0Fu) | wmask;
      const int off = j * 4096;
      e[0] = lds_read_u32x2(__builtin_amdgcn_perm(lo, lane_off, 0x0C010400u) + off);
      e[1] = lds_read_u32x2(__builtin_amdgcn_perm(hi, lane_off, 0x0C010400u) + off);
      e[2] = lds_read_u32x2(__builtin_amdgcn_perm(lo, lane_off, 0x0C010500u) + off);
      e[3] = lds_read_u32x2(__builtin_amdgcn_perm(hi, lane_off, 0x0C010500u) + off);
      e[4] = lds_read_u32x2(__builtin_amdgcn_perm(lo, lane_off, 0x0C010600u) + off);
      e[5] = lds_read_u32x2(__builtin_amdgcn_perm(hi, lane_off, 0x0C010600u) + off);
      e[6] = lds_read_u32x2(__builtin_amdgcn_perm(lo, lane_off, 0x0C010700u) + off);
      e[7] = lds_read_u32x2(__builtin_amdgcn_perm(hi, lane_off, 0x0C010700u) + off);
    } else {
      const uint32_t tbj = j * 2048 + lane_off;
      e[0] = lds_read_u32x2(tbj | field3_x256<8 * PH + 0>(t0[j], t1[j], t2[j]));
      e[1] = lds_read_u32x2(tbj | field3_x256<8 * PH + 1>(t0[j], t1[j], t2[j]));
      e[2] = lds_read_u32x2(tbj | field3_x256<8 * PH + 2>(t0[j], t1[j], t2[j]));
      e[3] = lds_read_u32x2(tbj | field3_x256<8 * PH + 3>(t0[j], t1[j], t2[j]));
      e[4] = lds_read_u32x2(tbj | field3_x256<8 * PH + 4>(t0[j], t1[j], t2[j]));
      e[5] = lds_read_u32x2(tbj | field3_x256<8 * PH + 5>(t0[j], t1[j], t2[j]));
      e[6] = lds_read_u32x2(tbj | field3_x256<8 * PH + 6>(t0[j], t1[j], t2[j]));
      e[7] = lds_read_u32x2(tbj | field3_x256<8 * PH + 7>(t0[j], t1[j], t2[j]));
    }
    uint32_t bh[4], bm[4], bl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bh[i] = __builtin_amdgcn_perm(e[2 * i + 1].x, e[2 * i].x, 0x05040100u);  // the low halves: hi parts
      bm[i] = __builtin_amdgcn_perm(e[2 * i + 1].x, e[2 * i].x, 0x07060302u);  // the high halves: mid parts
      bl[i] = __builtin_amdgcn_perm(e[2 * i + 1].y, e[2 * i].y, 0x05040100u);
    }
    const bf16x8 Bh = as_frag(bh), Bm = as_frag(bm), Bl = as_frag(bl);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const bf16x8 Ah = as_frag(ah[mb]), Am = as_frag(am[mb]), Al = as_frag(al[mb]);
      f32x4 c = acc[mb][j];
      // small partial products first
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, Bm, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bl, c, 0, 0, 0);
      if constexpr (XMODE != 2) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al, Bh, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bm, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am, Bh, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah, Bh, c, 0, 0, 0);
      acc[mb][j] = c;
    }
  }
}

// ---- non-finite operands ----
// The split is exact for finite values only: an infinite vec value or codebook entry gives inf - inf = NaN in its mid part,
// and every sum it takes part in comes out NaN, where the reference's fp32 FMA chain (squeezellm/quant_cuda_kernel.cu:
// 1011-1036) gives +-inf (or NaN only for inf - inf and 0 x inf).  A finite result therefore proves finite operands
// (and an overflowing sum is non-finite in both); a NON-FINITE sum of the matrix instructions is recomputed here the
// reference's way -- one fp32 FMA chain over the workgroup's k's, index by index out of qweight -- before it is added to
// mul.  Cold code: it runs for outputs that NaN / inf operands reach, at ~10 instructions per weight.
__device__ __forceinline__ bool is_finite_f32(float v) { return (__builtin_bit_cast(uint32_t, v) & 0x7F800000u) != 0x7F800000u; }

// index of weight (k, col) in the packed matrix (squeezellm/quant.py:180-203: 4-bit -- nibble k % 8 of row k / 8;
// 3-bit -- field k % 32 of the 96-bit little-endian stream in rows 3 (k / 32) .. + 2)
template <int BITS>
__device__ __forceinline__ uint32_t packed_index(const uint32_t* __restrict__ q, int N, int k, int col) {
  if constexpr (BITS == 4) {
    return (q[(size_t)(k >> 3) * N + col] >> (4 * (k & 7))) & 15u;
  } else {
    const int bit = 3 * (k & 31), w = bit >> 5, o = bit & 31;
    const size_t base = (size_t)(3 * (k >> 5) + w) * N + col;
    uint32_t f = q[base] >> o;
    if (o > 29) f |= q[base + N] << (32 - o);
    return f & 7u;
  }
}

// sum over k in [k_beg, k_end) of lookup_table[col][index(k, col)] * vec[row][k] as ONE fp32 FMA chain
template <int BITS>
__device__ __forceinline__ float dense_term_fp32(const float* __restrict__ x_row, const uint32_t* __restrict__ q,
                                                 const float* __restrict__ lut, int N, int col, int k_beg, int k_end) {
  const float* lc = lut + (size_t)col * (1 << BITS);
  float s = 0.f;
  for (int k = k_beg; k < k_end; ++k) s = __builtin_fmaf(lc[packed_index<BITS>(q, N, k, col)], x_row[k], s);
  return s;
}

// exact split of one codebook value into its LDS entry {hi | mid << 16, lo}
__device__ __forceinline__ u32x2 split_entry(float v) {
  const uint32_t b = __builtin_bit_cast(uint32_t, v);
  const uint32_t hb = b & 0xFFFF0000u;
  const float r1 = v - __builtin_bit_cast(float, hb);
  const uint32_t mbits = __builtin_bit_cast(uint32_t, r1) & 0xFFFF0000u;
  const float r2 = r1 - __builtin_bit_cast(float, mbits);
  return u32x2{(hb >> 16) | mbits, __builtin_bit_cast(uint32_t, r2) >> 16};
}

}  // namespace sqllm
