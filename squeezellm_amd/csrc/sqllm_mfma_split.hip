// sqllm_mfma_split.hip -- wide-batch dense term on the bf16 MATRIX cores with fp32-class results: the *_batched
// operators from `mfma_min_batch` rows up (reference: one weight pass PER BATCH ROW, squeezellm/quant_cuda_kernel.cu:
// 884-979 / :982-1038).
//
// The fp32 matrix instruction (v_mfma_f32_16x16x4_f32, sqllm_kernels.hip: dense_role_mfma) runs at the fp32 VECTOR
// rate -- 1/16 of the bf16 matrix rate (MI355X_MICROARCH.md: 155 vs 2075-2382 TFLOP/s measured) -- so a wide batch was
// bound by the matrix pipe: 13B gate/up 30 us for 9..16 rows, 2048 rows 2.4 ms (120 of 157 TFLOP/s).  Here every fp32
// operand is written as the EXACT sum of three bf16 values,
//         v = hi + mid + lo,   hi = top 8 significant bits of v, mid = the next 8, lo = the last 8
// (truncations of v, v - hi, v - hi - mid: each difference is exact in fp32), and the product w * x as the six
// partial products whose magnitude can reach 2^-16 of it or more:
//         w x  ~  wh xh + wh xm + wm xh + wh xl + wl xh + wm xm            (dropped: wm xl, wl xm, wl xl <= 2^-24 |w x|)
// A bf16 x bf16 product is exact in fp32, the matrix instruction accumulates in fp32, so the result carries the
// rounding of an fp32 FMA chain plus 3 x 2^-24 per product: fp32 class (measured against the fp64 oracle next to the
// fp32 kernel: tests/test_gpu_batched.py, same 2e-5 gate, observed ~3e-7).  Six v_mfma_f32_16x16x32_bf16 (32 k's
// each, 16 cycles) replace 8 x v_mfma_f32_16x16x4_f32 (4 k's each, 32 cycles): 96 instead of 256 matrix-pipe cycles
// per 16 rows x 16 columns x 32 k's.
//
// Where the splits come from:
//   * weights: the tile's codebook is staged in LDS ALREADY SPLIT -- an 8-byte entry {hi | mid << 16, lo} per
//     (column, index), layout [column j of the lane's four][index][32 slots x 8 B] (two copies of the 16 column
//     groups side by side: a half-wave's ds_read_b64 then touches 32 different 8-byte slots of one 256-byte row,
//     conflict-free whatever the indices are); a lookup is ONE ds_read_b64 per weight, addressed by one
//     v_perm_b32 as in the batch-1 kernel, and three v_perm_b32 per TWO weights pack the halves into the operand
//     registers;
//   * vec: lane (c, kq) of the wave reads the 8 consecutive k's of batch row c that belong to ITS qweight row
//     straight from global memory (32 bytes; the four lane rows cover one 128-byte line per batch row) and splits
//     them in registers: and / sub / and / sub per value + the same packing.
// Operand mapping: lane (c = l % 16, kq = l / 16) holds nibbles 0..7 of its own packed word = k = 8 (r0 + kq) + s as
// the 8 elements of its B fragment, and x[row c][8 (r0 + kq) + s] as its A fragment: both sides enumerate the k's
// of a matrix instruction the same way, which is all it needs (no cross-lane traffic; cf. dense_role_mfma).
// Work decomposition, prefetch ping-pong, LDS meeting of the waves and the epilogue are the fp32 kernel's.
//
// The WIDE form (64 rows and more, sqllm_fused_wide below) takes vec split ONCE, by its own kernel (sqllm_split_vec), into
// bf16 planes in stream-ordered scratch, laid out in FRAGMENT order: a 1-KB block per (16 rows, 32 k's, plane) holds the
// 64 lanes' 16-byte A fragments back to back, blocks ordered [row block][k block][plane hi, mid, lo] -- a wave's load is
// 1 KB contiguous, and no value is split more than once (in the kernels above: once per 64-column tile).  Every row
// block ends in an all-zero k block: the address of lane rows past the end of a K range.  The split
// kernel also reports whether any `lo` part is non-zero; where none is (vec came from fp16 values, as in
// QuantLinearLUT.forward: 11 significant bits fit hi + mid) the lo plane is neither read nor multiplied (five partial
// products instead of six).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <type_traits>

#include "sqllm_kernels.h"

#include "sqllm_decode.h"
#include "sqllm_roles.h"

namespace sqllm {

namespace {

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

// exact split of one codebook value into its LDS entry {hi | mid << 16, lo}
__device__ __forceinline__ u32x2 split_entry(float v) {
  const uint32_t b = __builtin_bit_cast(uint32_t, v);
  const uint32_t hb = b & 0xFFFF0000u;
  const float r1 = v - __builtin_bit_cast(float, hb);
  const uint32_t mbits = __builtin_bit_cast(uint32_t, r1) & 0xFFFF0000u;
  const float r2 = r1 - __builtin_bit_cast(float, mbits);
  return u32x2{(hb >> 16) | mbits, __builtin_bit_cast(uint32_t, r2) >> 16};
}

template <int BITS, int MB, int WAVES>
__device__ __forceinline__ void dense_role_mfma_split(const float* __restrict__ x, const u32x4* __restrict__ q,
                                                      float* __restrict__ y, const float* __restrict__ lut, int K, int N,
                                                      int batch, int m0, int bid, int n_col_tiles, int units_total,
                                                      int units_per_wg, int units_stride, float* lds) {
  constexpr int XMODE = 0;  // vec = fp32 rows, split in registers
  constexpr int NX = 2 * MB;  // 16-byte registers of one phase's vec values
  using F = Fmt<BITS>;
  constexpr int L = F::kLut, R = F::kRows, KU = F::kK;
  constexpr int NPH = KU / 8;  // phases of 8 k's per unit (4-bit: 1, 3-bit: 4)
  constexpr int T = WAVES * 64;
  __builtin_amdgcn_s_waitcnt(0);  // clean slate for the compiler's wait-count model (see dense_role)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, grp = lane >> 4;
  constexpr int kCbBytes = split_codebook_bytes(BITS);
  float* slabs = lds + kCbBytes / 4;
  const int row_stride = N / 4;  // in 16-byte units
  const char* qbase = reinterpret_cast<const char*>(q);
  const uint32_t row_bytes = 16u * (uint32_t)row_stride;
  // this lane's batch rows: row i16 of every block of 16 (rows past the batch re-read its last row; never stored)
  int xrow[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    int r = m0 + 16 * mb + i16;
    if (r > batch - 1) r = batch - 1;
    xrow[mb] = r * K;
  }
  const uint32_t lane_off = 8 * (i16 + 16 * (grp & 1));  // byte offset of this lane's slot inside an entry row

  const unsigned total = (unsigned)n_col_tiles * (unsigned)units_stride;
  unsigned gpos = (unsigned)bid * (unsigned)units_per_wg;
  unsigned gend = gpos + (unsigned)units_per_wg;
  if (gend > total) gend = total;
  while (gpos < gend) {
  const int ct = (int)(gpos / (unsigned)units_stride);
  const int u_beg = (int)(gpos - (unsigned)ct * (unsigned)units_stride);
  if (u_beg >= units_total) { gpos = (unsigned)(ct + 1) * (unsigned)units_stride; continue; }  // (padding behind a tile's last range)
  int u_end = units_total;
  if ((unsigned)(u_end - u_beg) > gend - gpos) u_end = u_beg + (int)(gend - gpos);
  gpos += (unsigned)(u_end - u_beg);
  if (u_end == units_total) gpos = (unsigned)(ct + 1) * (unsigned)units_stride;  // skip the padding
  const int col0 = ct * kTileN;

  // ---- codebook values this thread stages: entry e = tid + T i of the tile's 4 * L * 32 eight-byte entries;
  //      row = e / 32 = (column j, index), slot = e % 32 = (copy, column group) ----
  constexpr int NST = 4 * L * 32 / T;
  float ev[NST];
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    const int e = tid + T * i;
    const int row = e >> 5, slot = e & 31;
    int c = col0 + 4 * (slot & 15) + row / L;
    if (c > N - 1) c = N - 1;
    ev[i] = lut[(size_t)c * L + (row % L)];
  }
  const int n_groups_wg = (u_end - u_beg + 3) / 4;
  const int n_g = n_groups_wg > wave ? (n_groups_wg - wave + WAVES - 1) / WAVES : 0;
  int cidx = col0 / 4 + i16;
  if (cidx > row_stride - 1) cidx = row_stride - 1;
  const uint32_t lane_bytes = 16u * (uint32_t)cidx;
  auto group_unit = [&](int g) {  // unit of this lane's row in the wave's group g (may be >= u_end)
    return u_beg + 4 * (wave + WAVES * g) + grp;
  };
  auto clamp_unit = [&](int u) {
    if (u > u_end - 1) u = u_end - 1;  // clamped re-read inside the slice; its x values are zeroed
    if (u < u_beg) u = u_beg;
    return u;
  };
  auto load_w = [&](int g, u32x4 (&dw)[R]) {
    const int u = clamp_unit(group_unit(g));
    const uint32_t off = (uint32_t)(u * R) * row_bytes + lane_bytes;
#pragma unroll
    for (int r = 0; r < R; ++r) dw[r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(qbase + (off + r * row_bytes)));
  };
  auto load_x = [&](int g, int ph, u32x4 (&dx)[NX]) {
    const int u = clamp_unit(group_unit(g));
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const float* p = x + xrow[mb] + u * KU + 8 * ph;
      dx[2 * mb] = *reinterpret_cast<const u32x4*>(p);
      dx[2 * mb + 1] = *reinterpret_cast<const u32x4*>(p + 4);
    }
  };
  u32x4 wa[R], wb[R];
  u32x4 xa[NX], xb[NX];
  load_w(0, wa);
  load_x(0, 0, xa);
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  // ---- stage the codebooks, split ----
  {
    char* base = reinterpret_cast<char*>(lds);
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      *reinterpret_cast<u32x2*>(base + 8 * (tid + T * i)) = split_entry(ev[i]);
    }
  }
  f32x4 acc[MB][4];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[mb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();  // codebooks staged (and everybody has left the previous piece's slabs)

  auto phase = [&](const u32x4 (&t)[R], auto ph_tag, const u32x4 (&dx)[NX], int g) {
    split_phase<BITS, MB, XMODE, decltype(ph_tag)::value>(t, dx, group_unit(g) < u_end, lane_off, 0u, acc);
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using P2 = std::integral_constant<int, 2>;
  using P3 = std::integral_constant<int, 3>;
  // decode group g out of (w, xcur = its phase-0 vec values); the NEXT group's weights (into wn) are loaded before the
  // first phase, its phase-0 values (into xn) before the last.  With four phases the values of phase p + 1 are loaded
  // while phase p runs, alternating between xcur and xo -- the next group's phase 0 lands in xcur again (xn == xcur).
  auto decode_group = [&](int g, const u32x4 (&w)[R], u32x4 (&xcur)[NX], u32x4 (&wn)[R], u32x4 (&xn)[NX], u32x4 (&xo)[NX]) {
    load_w(g + 1, wn);
    if constexpr (NPH == 1) {
      load_x(g + 1, 0, xn);
      __builtin_amdgcn_sched_barrier(0);
      phase(w, P0{}, xcur, g);
    } else {
      load_x(g, 1, xo);
      __builtin_amdgcn_sched_barrier(0);
      phase(w, P0{}, xcur, g);
      load_x(g, 2, xcur);
      __builtin_amdgcn_sched_barrier(0);
      phase(w, P1{}, xo, g);
      load_x(g, 3, xo);
      __builtin_amdgcn_sched_barrier(0);
      phase(w, P2{}, xcur, g);
      load_x(g + 1, 0, xn);
      __builtin_amdgcn_sched_barrier(0);
      phase(w, P3{}, xo, g);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int g = 0; g < n_g; g += 2) {
    if constexpr (NPH == 1) {
      decode_group(g, wa, xa, wb, xb, xb);
      decode_group(g + 1, wb, xb, wa, xa, xa);
    } else {
      decode_group(g, wa, xa, wb, xa, xb);
      decode_group(g + 1, wb, xa, wa, xa, xb);
    }
  }

  // ---- waves meet in LDS, one row block at a time (see dense_role_mfma) ----
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


// ---- the wide form's phase: software-pipelined by hand ----
// LDS byte addresses of the 8 weights of column J (0..3 of the lane's four) in phase PH of the unit whose words are t;
// the column's table offset (J * kColStride) goes into the read's immediate field.  lane_off / wmask: see split_phase.
template <int BITS>
constexpr int kColStride = BITS == 4 ? 4096 : 2048;

template <int BITS, int PH>
__device__ __forceinline__ void col_addrs(const u32x4 (&t)[Fmt<BITS>::kRows], int J, uint32_t lane_off, uint32_t wmask, uint32_t (&a)[8]) {
  if constexpr (BITS == 4) {
    const uint32_t w = t[0][J];
    const uint32_t lo = (w & 0x0F0F0F0Fu) | wmask, hi = ((w >> 4) & 0x0F0F0F0Fu) | wmask;
    a[0] = __builtin_amdgcn_perm(lo, lane_off, 0x0C010400u);
    a[1] = __builtin_amdgcn_perm(hi, lane_off, 0x0C010400u);
    a[2] = __builtin_amdgcn_perm(lo, lane_off, 0x0C010500u);
    a[3] = __builtin_amdgcn_perm(hi, lane_off, 0x0C010500u);
    a[4] = __builtin_amdgcn_perm(lo, lane_off, 0x0C010600u);
    a[5] = __builtin_amdgcn_perm(hi, lane_off, 0x0C010600u);
    a[6] = __builtin_amdgcn_perm(lo, lane_off, 0x0C010700u);
    a[7] = __builtin_amdgcn_perm(hi, lane_off, 0x0C010700u);
  } else {
    const uint32_t t0 = t[0][J], t1 = t[1][J], t2 = t[2][J];
    a[0] = lane_off | field3_x256<8 * PH + 0>(t0, t1, t2);
    a[1] = lane_off | field3_x256<8 * PH + 1>(t0, t1, t2);
    a[2] = lane_off | field3_x256<8 * PH + 2>(t0, t1, t2);
    a[3] = lane_off | field3_x256<8 * PH + 3>(t0, t1, t2);
    a[4] = lane_off | field3_x256<8 * PH + 4>(t0, t1, t2);
    a[5] = lane_off | field3_x256<8 * PH + 5>(t0, t1, t2);
    a[6] = lane_off | field3_x256<8 * PH + 6>(t0, t1, t2);
    a[7] = lane_off | field3_x256<8 * PH + 7>(t0, t1, t2);
  }
}

// ... the same one address at a time (i = 0..7, a constant once the caller's loops are unrolled): `pre` = the column's
// words prepared by col_prep -- 4-bit {low nibbles | wmask, high nibbles | wmask}, 3-bit the unit's three words
template <int BITS>
__device__ __forceinline__ void col_prep(const u32x4 (&t)[Fmt<BITS>::kRows], int J, uint32_t wmask, uint32_t (&pre)[3]) {
  if constexpr (BITS == 4) {
    const uint32_t w = t[0][J];
    pre[0] = (w & 0x0F0F0F0Fu) | wmask;
    pre[1] = ((w >> 4) & 0x0F0F0F0Fu) | wmask;
    pre[2] = 0;
  } else {
    pre[0] = t[0][J]; pre[1] = t[1][J]; pre[2] = t[2][J];
  }
}
template <int BITS, int PH>
__device__ __forceinline__ uint32_t col_addr(const uint32_t (&pre)[3], int i, uint32_t lane_off) {
  if constexpr (BITS == 4) {
    const uint32_t src = pre[i & 1];
    switch (i >> 1) {
      case 0: return __builtin_amdgcn_perm(src, lane_off, 0x0C010400u);
      case 1: return __builtin_amdgcn_perm(src, lane_off, 0x0C010500u);
      case 2: return __builtin_amdgcn_perm(src, lane_off, 0x0C010600u);
      default: return __builtin_amdgcn_perm(src, lane_off, 0x0C010700u);
    }
  } else {
    switch (i) {
      case 0: return lane_off | field3_x256<8 * PH + 0>(pre[0], pre[1], pre[2]);
      case 1: return lane_off | field3_x256<8 * PH + 1>(pre[0], pre[1], pre[2]);
      case 2: return lane_off | field3_x256<8 * PH + 2>(pre[0], pre[1], pre[2]);
      case 3: return lane_off | field3_x256<8 * PH + 3>(pre[0], pre[1], pre[2]);
      case 4: return lane_off | field3_x256<8 * PH + 4>(pre[0], pre[1], pre[2]);
      case 5: return lane_off | field3_x256<8 * PH + 5>(pre[0], pre[1], pre[2]);
      case 6: return lane_off | field3_x256<8 * PH + 6>(pre[0], pre[1], pre[2]);
      default: return lane_off | field3_x256<8 * PH + 7>(pre[0], pre[1], pre[2]);
    }
  }
}

// packed B operand word k of 12 ({hi x 4, mid x 4, lo x 4}) out of the 8 looked-up entries
__device__ __forceinline__ uint32_t pack_b(const u32x2 (&e)[8], int k) {
  const int i = k & 3, kind = k >> 2;
  return kind == 0   ? __builtin_amdgcn_perm(e[2 * i + 1].x, e[2 * i].x, 0x05040100u)   // the low halves: hi parts
         : kind == 1 ? __builtin_amdgcn_perm(e[2 * i + 1].x, e[2 * i].x, 0x07060302u)   // the high halves: mid parts
                     : __builtin_amdgcn_perm(e[2 * i + 1].y, e[2 * i].y, 0x05040100u);
}

// One phase (8 k's of each lane row x the lane's four columns x 64 rows) of the wide form.  In split_phase a wave
// alternates between looking a column up (addresses, 8 LDS reads, their latency, 12 packing instructions) and the 20-24
// matrix instructions that use it -- with two waves per SIMD the matrix pipe was 65 % busy (profiles/r04_wide_pmc.txt).
// Here the lookups of the NEXT column ride between the matrix instructions of the current one, slot by slot (a
// scheduling barrier after each keeps the order): slot 0 its index words, slots 2-5 two addresses + reads each, slots 8-19 one packing
// instruction each.  B enters holding the packed operands of (t, PH, column 0) and leaves holding those of
// (tn, PHN, column 0), the first column of the phase that follows.
template <int BITS, int XMODE, int PH, int PHN>
__device__ __forceinline__ void wide_phase(const u32x4 (&t)[Fmt<BITS>::kRows], const u32x4 (&tn)[Fmt<BITS>::kRows],
                                           const u32x4 (&dx)[XMODE == 0 ? 8 : XMODE * 4], bool live, uint32_t lane_off, uint32_t wmask,
                                           uint32_t (&B)[12], f32x4 (&acc)[4][4]) {
  constexpr int MB = 4;
  constexpr int NP = XMODE == 2 ? 5 : 6;  // partial products
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
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t pre[3], Bn[12];
    u32x2 e[8];
    const uint32_t bh[4] = {B[0], B[1], B[2], B[3]}, bm[4] = {B[4], B[5], B[6], B[7]}, bl[4] = {B[8], B[9], B[10], B[11]};
    const bf16x8 Bh = as_frag(bh), Bm = as_frag(bm), Bl = as_frag(bl);
    const int jn = j < 3 ? j + 1 : 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int slot = MB * p + mb;
        const bf16x8 Ah = as_frag(ah[mb]), Am = as_frag(am[mb]), Al = as_frag(al[mb]);
        // small partial products first: Am Bm, Ah Bl, [Al Bh,] Ah Bm, Am Bh, Ah Bh
        const int pp = (XMODE == 2 && p >= 2) ? p + 1 : p;
        const bf16x8 A = (pp == 0 || pp == 4) ? Am : pp == 2 ? Al : Ah;
        const bf16x8 Bx = (pp == 0 || pp == 3) ? Bm : pp == 1 ? Bl : Bh;
        acc[mb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, Bx, acc[mb][j], 0, 0, 0);
        if (slot == 0) {
          if (j < 3) col_prep<BITS>(t, jn, wmask, pre);
          else col_prep<BITS>(tn, 0, wmask, pre);
        }
        if (slot >= 2 && slot < 6) {  // (addresses where they are used: eight of them alive at once cost the 3-bit kernel its last registers)
#pragma unroll
          for (int i = 2 * (slot - 2); i < 2 * (slot - 2) + 2; ++i) {
            const uint32_t ad = j < 3 ? col_addr<BITS, PH>(pre, i, lane_off) : col_addr<BITS, PHN>(pre, i, lane_off);
            e[i] = lds_read_u32x2(ad + jn * kColStride<BITS>);
          }
        }
        if (slot >= 8 && slot < 20) {
          // (order: the words that need the earliest reads first -- pair 0's hi, mid, lo, then pair 1's ...)
          const int k = slot - 8, i = k / 3, kind = k % 3;
          Bn[4 * kind + i] = pack_b(e, 4 * kind + i);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) B[k] = Bn[k];
  }
}

// packed operands of (t, PH, column 0) from scratch (a workgroup's first phase)
template <int BITS, int PH>
__device__ __forceinline__ void first_column(const u32x4 (&t)[Fmt<BITS>::kRows], uint32_t lane_off, uint32_t wmask, uint32_t (&B)[12]) {
  uint32_t a[8];
  u32x2 e[8];
  col_addrs<BITS, PH>(t, 0, lane_off, wmask, a);
#pragma unroll
  for (int i = 0; i < 8; ++i) e[i] = lds_read_u32x2(a[i]);
#pragma unroll
  for (int k = 0; k < 12; ++k) B[k] = pack_b(e, k);
}

// ------------------------------------------------------------------------------------------------
// 64 rows and more: the WIDE form.  The kernels above give a workgroup ONE 64-column tile and divide its k's between
// the eight waves: every wave reads different vec values, nothing a wave loads is of use to another, and at 2048
// rows the 64 x K values of a row block are fetched once per column tile -- 9 GB per 13B gate/up op, with 9 % of it
// found in the L2 (profiles/r04_split_planes_pmc.txt: 32 workgroups per XCD on 32 different row blocks), i.e. a kernel
// bound by the fabric at 5.8 TB/s with the matrix pipe 43 % busy.  Here a workgroup takes EIGHT column tiles -- one
// per wave, codebook table private to the wave, 16 KB (4-bit) / 8 KB (3-bit) of LDS each -- and every wave walks the
// SAME k's: the A fragments of a step are fetched from the L2 once per workgroup and found in the CU's vector cache by
// the other seven waves (eight times less vec traffic), no cross-wave sum, no barrier anywhere in the kernel; a wave
// adds its 64 x 64 results straight to mul.  Grid: x = (group of 8 column tiles, K slice), y = block of 64 rows.
// ------------------------------------------------------------------------------------------------
constexpr int kWideTiles = 8;  // column tiles per workgroup = waves
constexpr int kWideSlabFloats = 64 * kTileN;  // a wave's 64 x 64 sums

template <int BITS, int XMODE>
__device__ __forceinline__ void dense_role_mfma_wide(const void* __restrict__ xv, const u32x4* __restrict__ q,
                                                     float* __restrict__ y, const float* __restrict__ lut, int K, int N, int batch,
                                                     int m0, int ct, int u_beg, int u_end, bool atomic, float* __restrict__ slab) {
  using F = Fmt<BITS>;
  constexpr int MB = 4;
  constexpr int L = F::kLut, R = F::kRows, KU = F::kK;
  constexpr int NPH = KU / 8;
  constexpr int NX = XMODE == 0 ? 2 * MB : XMODE * MB;
  constexpr int kCbBytes = split_codebook_bytes(BITS);
  const float* x = static_cast<const float*>(xv);
  const uint32_t KB = (uint32_t)K / 32;
  __builtin_amdgcn_s_waitcnt(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, grp = lane >> 4;
  const int row_stride = N / 4;  // in 16-byte units
  const char* qbase = reinterpret_cast<const char*>(q);
  const uint32_t row_bytes = 16u * (uint32_t)row_stride;
  int xrow[MB];        // fp32 vec: this lane's batch rows (rows past the batch re-read its last row; never stored)
  const char* xblk[MB];   // planes: the row block's k block 0 (wave-uniform: a scalar base for the loads)
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    int r = m0 + 16 * mb + i16;
    if (r > batch - 1) r = batch - 1;
    xrow[mb] = r * K;
    xblk[mb] = static_cast<const char*>(xv) + (size_t)(m0 / 16 + mb) * (KB + 1) * 3072;
  }
  const int col0 = ct * kTileN;
  const uint32_t wbase = (uint32_t)wave * (uint32_t)kCbBytes;  // this wave's table
  const uint32_t slot = 8 * (i16 + 16 * (grp & 1));
  // (see split_phase: a 4-bit table base is 16 KB * wave -- bits 14, 15 ride in the index bytes, bit 16 in byte 1 of lane_off)
  const uint32_t lane_off = BITS == 4 ? (slot | ((wbase >> 16) << 8)) : (wbase + slot);
  uint32_t wmask = BITS == 4 ? 0x01010101u * ((wbase >> 8) & 0xC0u) : 0u;
  asm volatile("" : "+v"(wmask));  // (in a vector register: (w & 0x0F0F0F0F) | wmask is then ONE v_and_or_b32 -- two scalar operands would not encode)

  // ---- this wave's codebook: lane = column of the tile, its L values as L / 4 sixteen-byte loads (the tile's block of
  // the table is contiguous: 64 columns x L floats).  Gathering entry by entry in LDS order -- 32 scattered loads per
  // lane, as the tile kernels do with 512 threads -- cost 92 us of a 1.03-ms launch at 2048 rows: every workgroup of a
  // round builds its tables at the same moment (profiles/r04_wide_ablate_midrows.txt, noLut). ----
  constexpr int NLV = L / 4;
  f32x4 lv[NLV];
  {
    int c = col0 + lane;
    if (c > N - 1) c = N - 1;  // (columns past N: the last one again; never stored)
    const f32x4* lp = reinterpret_cast<const f32x4*>(lut + (size_t)c * L);
#pragma unroll
    for (int i = 0; i < NLV; ++i) lv[i] = lp[i];
  }
  const int n_g = (u_end - u_beg + 3) / 4;
  int cidx = col0 / 4 + i16;
  if (cidx > row_stride - 1) cidx = row_stride - 1;
  const uint32_t lane_bytes = 16u * (uint32_t)cidx;
  auto group_unit = [&](int g) { return u_beg + 4 * g + grp; };
  auto clamp_unit = [&](int u) {
    if (u > u_end - 1) u = u_end - 1;
    return u;
  };
  auto load_w = [&](int g, u32x4 (&dw)[R]) {
    const int u = clamp_unit(group_unit(g));
    const uint32_t off = (uint32_t)(u * R) * row_bytes + lane_bytes;
#pragma unroll
    for (int r = 0; r < R; ++r) dw[r] = *reinterpret_cast<const u32x4*>(qbase + (off + r * row_bytes));  // (re-read by every row block: cached)
  };
  auto load_x = [&](int g, int ph, u32x4 (&dx)[NX]) {
    const int gu = group_unit(g);
    const int u = clamp_unit(gu);
    if constexpr (XMODE == 0) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const float* p = x + xrow[mb] + u * KU + 8 * ph;
        dx[2 * mb] = *reinterpret_cast<const u32x4*>(p);
        dx[2 * mb + 1] = *reinterpret_cast<const u32x4*>(p + 4);
      }
    } else {
      // fragment order (sqllm_split_vec): 4-bit -- the group's 32 k's are ONE k block, lane for lane; 3-bit -- a lane
      // row's unit is a k block of its own, phase ph = its quarter.  One 32-bit byte offset serves all row blocks and
      // planes (scalar base per row block, the plane in the immediate field); past the K range: the zero k block.
      const uint32_t kb = gu < u_end ? (BITS == 4 ? (uint32_t)u >> 2 : (uint32_t)u) : KB;
      const uint32_t lp = BITS == 4 ? (uint32_t)lane : (uint32_t)(16 * ph + i16);
      const uint32_t off = 3072u * kb + 16u * lp;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
        for (int pl = 0; pl < XMODE; ++pl) dx[XMODE * mb + pl] = *reinterpret_cast<const u32x4*>(xblk[mb] + off + 1024 * pl);
      }
    }
  };
  // weights two groups ahead (they come from HBM / the Infinity Cache), vec values one phase ahead (L2 / vector cache)
  u32x4 wa[R], wb[R];
  u32x4 xa[NX], xb[NX];
  load_w(0, wa);
  load_w(1, wb);
  load_x(0, 0, xa);
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  {
    // entry (column c, index) lives at [c % 4][index][slot c / 4, and again at slot 16 + c / 4] (see the header)
    const uint32_t ebase = wbase + (uint32_t)(lane & 3) * (uint32_t)(L * 256) + 8u * (uint32_t)(lane >> 2);
#pragma unroll
    for (int i = 0; i < NLV; ++i) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const u32x2 en = split_entry(lv[i][t]);
        typedef u32x2 __attribute__((address_space(3))) lds_u32x2;
        *reinterpret_cast<lds_u32x2*>(ebase + 256u * (uint32_t)(4 * i + t)) = en;
        *reinterpret_cast<lds_u32x2*>(ebase + 256u * (uint32_t)(4 * i + t) + 128u) = en;
      }
    }
  }
  f32x4 acc[MB][4];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[mb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the table is this wave's own: LDS operations of one wave complete in order, no barrier
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // group g out of (w = its words, wn = the next group's, xcur = its phase-0 values); w is refilled with group g + 2's
  // words once a copy is taken (4-bit) or the group is through (3-bit).  Loads return in order: the vec loads of the next phase go out BEFORE the far-ahead
  // weight load, so that waiting for them does not mean waiting for it.
  uint32_t Bst[12];  // packed B operands of the upcoming phase's first column (wide_phase)
  first_column<BITS, 0>(wa, lane_off, wmask, Bst);
  auto decode_group = [&](int g, u32x4 (&w)[R], const u32x4 (&wn)[R], u32x4 (&xcur)[NX], u32x4 (&xn)[NX], u32x4 (&xo)[NX]) {
    const bool live = group_unit(g) < u_end;
    if constexpr (NPH == 1) {
      u32x4 t[R];
#pragma unroll
      for (int r = 0; r < R; ++r) t[r] = w[r];
      load_x(g + 1, 0, xn);
      load_w(g + 2, w);
      __builtin_amdgcn_sched_barrier(0);
      wide_phase<BITS, XMODE, 0, 0>(t, wn, xcur, live, lane_off, wmask, Bst, acc);
    } else {
      // (a group is four phases long: its words are refilled with group g + 2's once its last phase is through -- no copy)
      load_x(g, 1, xo);
      __builtin_amdgcn_sched_barrier(0);
      wide_phase<BITS, XMODE, 0, 1>(w, w, xcur, live, lane_off, wmask, Bst, acc);
      load_x(g, 2, xcur);
      __builtin_amdgcn_sched_barrier(0);
      wide_phase<BITS, XMODE, 1, 2>(w, w, xo, live, lane_off, wmask, Bst, acc);
      load_x(g, 3, xo);
      __builtin_amdgcn_sched_barrier(0);
      wide_phase<BITS, XMODE, 2, 3>(w, w, xcur, live, lane_off, wmask, Bst, acc);
      load_x(g + 1, 0, xn);
      __builtin_amdgcn_sched_barrier(0);
      wide_phase<BITS, XMODE, 3, 0>(w, wn, xo, live, lane_off, wmask, Bst, acc);
      load_w(g + 2, w);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int g = 0; g < n_g; g += 2) {
    if constexpr (NPH == 1) {
      decode_group(g, wa, wb, xa, xb, xb);
      decode_group(g + 1, wb, wa, xb, xa, xa);
    } else {
      decode_group(g, wa, wb, xa, xa, xb);
      decode_group(g + 1, wb, wa, xa, xa, xb);
    }
  }
  // ---- results: lane (i16, grp) holds rows 16 mb + 4 grp + {x, y, z, w} of columns 4 i16 + j.  A workgroup that covered
  // all of K owns its outputs (the launch's other workgroups write other tiles, the sparse terms were an earlier launch):
  // 16-byte read-add-write; K slices add atomically -- the L2 takes ~1.2 fp32 atomics per clock and channel, 57 M of them
  // (13B gate/up, 2048 rows, two slices) were 177 us of a 1.39-ms kernel (profiles/r04_wide_ablate.txt).
  const int c0 = col0 + 4 * i16;
  if (slab) {
    // a K slice with scratch: its 64 x 64 sums go out as a 16-KB slab in lane order (16 stores of 1 KB per wave); the
    // launch that follows (sqllm_wide_reduce) adds a tile's slabs to mul.  Adding them here atomically cost 74 of 151 us
    // at 128 rows, 68 of 208 at 256 (profiles/r04_wide_ablate_midrows.txt).
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        *reinterpret_cast<f32x4*>(slab + (size_t)((mb * 4 + e) * 64 + lane) * 4) = f32x4{acc[mb][0][e], acc[mb][1][e], acc[mb][2][e], acc[mb][3][e]};
  } else if (c0 < N) {  // (N is a multiple of 4: the lane's four columns exist together)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int r0 = m0 + 16 * mb + 4 * grp;
      float* p = y + (size_t)r0 * N + c0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (r0 + e < batch) {
          float* pe = p + (size_t)e * N;
          const f32x4 v = {acc[mb][0][e], acc[mb][1][e], acc[mb][2][e], acc[mb][3][e]};
          if (atomic) {
            acc_add(pe + 0, v.x);
            acc_add(pe + 1, v.y);
            acc_add(pe + 2, v.z);
            acc_add(pe + 3, v.w);
          } else {
            f32x4 o = *reinterpret_cast<const f32x4*>(pe);
            o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
            *reinterpret_cast<f32x4*>(pe) = o;
          }
        }
      }
    }
  }
}
}  // namespace

template <int BITS, int MB, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, (MB == 1 || (BITS == 4 && MB == 2)) ? 4 : 2)
sqllm_fused_batched_split(const float* x, const GroupArgs ga) {
  __shared__ __attribute__((aligned(16))) float lds[split_lds_floats(BITS, WAVES)];
  const Segment sg = ga.seg[0];  // the whole descriptor in one round of scalar loads (see sqllm_fused_matvec)
  asm volatile("" ::SQLLM_SEG_OPERANDS(sg), "s"(x));
  __builtin_amdgcn_sched_barrier(0);
  const KernelGeom& gm = sg.gm;
  const int m0 = blockIdx.y * 16 * MB;
  dense_role_mfma_split<BITS, MB, WAVES>(x, reinterpret_cast<const u32x4*>(sg.q), sg.y, sg.lut, gm.K, gm.N, gm.batch, m0,
                                         (int)blockIdx.x, gm.col_tiles, gm.units_total, gm.units_per_wg,
                                         gm.dense_blocks == gm.col_tiles * gm.k_slices ? gm.k_slices * gm.units_per_wg : gm.units_total,
                                         lds);
}

// ------------------------------------------------------------------------------------------------
// vec split once into bf16 planes in fragment order (see the header).  Chunk = 16 bytes = 8 k's of one row; chunk index
//   ((rb * (K / 32 + 1) + kb) * 3 + plane) * 64 + lane,   lane = 16 * ((k / 8) % 4) + row % 16,  rb = row / 16, kb = k / 32,
// rows padded with zeros to a multiple of 64, k block K / 32 of every row block all zero.  flags[w] = 1 if workgroup w
// met a non-zero lo part; the grid is always kSplitFlagWgs workgroups, so the consumer ORs a fixed number of flags and
// nothing needs zeroing beforehand.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sqllm_split_vec(const float* __restrict__ x, u32x4* __restrict__ planes, uint32_t n_frag,
                                                       uint32_t* __restrict__ flags, int batch, int K) {
  const uint32_t KB1 = (uint32_t)K / 32 + 1;
  uint32_t any = 0;
  for (uint32_t f = blockIdx.x * 256 + threadIdx.x; f < n_frag; f += kSplitFlagWgs * 256) {  // f = (rb, kb, lane)
    const uint32_t lane = f & 63, blk = f >> 6;
    const uint32_t rb = blk / KB1, kb = blk - rb * KB1;
    const uint32_t row = 16 * rb + (lane & 15), k = 32 * kb + 8 * (lane >> 4);
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (row < (uint32_t)batch && kb + 1 < KB1) {
      const float* p = x + (size_t)row * K + k;
      const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    uint32_t h[4], m[4], l[4];
    split8(v, h, m, l);
    u32x4* o = planes + ((size_t)blk * 192 + lane);
    o[0] = u32x4{h[0], h[1], h[2], h[3]};
    o[64] = u32x4{m[0], m[1], m[2], m[3]};
    o[128] = u32x4{l[0], l[1], l[2], l[3]};
    any |= l[0] | l[1] | l[2] | l[3];
  }
  // (a lo part is 0 or a non-zero bf16 -- never -0: the differences above are exact, x - x = +0)
  const int wg_any = __syncthreads_or(any != 0);
  if (threadIdx.x == 0) flags[blockIdx.x] = wg_any ? 1u : 0u;
}

// wide form (dense_role_mfma_wide): XP = vec as bf16 planes (with the lo flags) or as fp32 rows (no scratch: split in registers).
// 1-D grid over UNITS = (64-row block rb, group of 8 column tiles cg), unit = rb * col_groups + cg: workgroups
// [0, full_units) take one unit each over all of K; the remaining units -- the last, partial round of one workgroup per
// CU -- are cut into gm.k_slices K slices of gm.units_per_wg units, one workgroup each (make_plan_wide).
template <int BITS, bool XP>
__global__ void __launch_bounds__(kWaves * 64, 2)
sqllm_fused_wide(const void* xv, const uint32_t* flags, int full_units, float* slabs, const GroupArgs ga) {
  __shared__ __attribute__((aligned(16))) char lds[kWideTiles * split_codebook_bytes(BITS)];
  static_assert(kWaves == kWideTiles, "one column tile per wave");
  const Segment sg = ga.seg[0];
  asm volatile("" ::SQLLM_SEG_OPERANDS(sg), "s"(xv), "s"(flags), "s"(full_units), "s"(slabs));
  __builtin_amdgcn_sched_barrier(0);
  const KernelGeom& gm = sg.gm;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  int unit = blockIdx.x, u_beg = 0, u_end = gm.units_total;
  const bool sliced = unit >= full_units && gm.k_slices > 1;
  float* slab = nullptr;
  if (sliced) {
    const int t = unit - full_units;
    const int q = t / gm.k_slices;
    unit = full_units + q;
    u_beg = (t - q * gm.k_slices) * gm.units_per_wg;
    if (u_end > u_beg + gm.units_per_wg) u_end = u_beg + gm.units_per_wg;
    if (slabs) slab = slabs + ((size_t)t * kWideTiles + wave) * kWideSlabFloats;  // [sliced unit][slice][wave]
  }
  const int col_groups = (gm.col_tiles + kWideTiles - 1) / kWideTiles;
  const int rb = unit / col_groups, cg = unit - rb * col_groups;
  const int ct = cg * kWideTiles + wave;
  asm volatile("" ::"v"(lds));  // (the role addresses the tables by number: keep the array)
  if (ct >= gm.col_tiles || u_beg >= u_end) return;  // (no barrier below: a wave without work just leaves)
  const int m0 = rb * 64;
  const u32x4* q = reinterpret_cast<const u32x4*>(sg.q);
  if constexpr (XP) {
    static_assert(kSplitFlagWgs == 256, "four flags per lane");
    const uint32_t f = flags[lane] | flags[lane + 64] | flags[lane + 128] | flags[lane + 192];
    const bool has_lo = __builtin_amdgcn_ballot_w64(f != 0) != 0;
    if (has_lo) dense_role_mfma_wide<BITS, 3>(xv, q, sg.y, sg.lut, gm.K, gm.N, gm.batch, m0, ct, u_beg, u_end, sliced, slab);
    else dense_role_mfma_wide<BITS, 2>(xv, q, sg.y, sg.lut, gm.K, gm.N, gm.batch, m0, ct, u_beg, u_end, sliced, slab);
  } else {
    dense_role_mfma_wide<BITS, 0>(xv, q, sg.y, sg.lut, gm.K, gm.N, gm.batch, m0, ct, u_beg, u_end, sliced, slab);
  }
}

// The K slices of the wide form's last round leave their sums as slabs (one per slice and wave = 64 x 64 tile, in the
// writing wave's lane order: element (16 mb + 4 grp + e, 4 i16 .. + 3) at float4 index (4 mb + e) * 64 + lane); this
// launch adds a tile's slabs to mul.  One workgroup of 256 threads per (sliced unit, wave).
__global__ void __launch_bounds__(256) sqllm_wide_reduce(const float* __restrict__ slabs, float* __restrict__ y, int N, int batch, int col_tiles,
                                                         int full_units, int k_slices) {
  const int uq = blockIdx.x / kWideTiles, w = blockIdx.x - uq * kWideTiles;
  const int col_groups = (col_tiles + kWideTiles - 1) / kWideTiles;
  const int unit = full_units + uq;
  const int rb = unit / col_groups, cg = unit - rb * col_groups;
  const int ct = cg * kWideTiles + w;
  if (ct >= col_tiles) return;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = threadIdx.x + 256 * i;  // float4 index inside a slab
    const int lane = idx & 63, me = idx >> 6;
    const int row = rb * 64 + 16 * (me >> 2) + 4 * (lane >> 4) + (me & 3), col = ct * kTileN + 4 * (lane & 15);
    f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int sl = 0; sl < k_slices; ++sl) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(slabs + (((size_t)uq * k_slices + sl) * kWideTiles + w) * kWideSlabFloats + (size_t)idx * 4);
      sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
    if (row < batch && col < N) {
      f32x4* p = reinterpret_cast<f32x4*>(y + (size_t)row * N + col);
      f32x4 o = *p;
      o.x += sum.x; o.y += sum.y; o.z += sum.z; o.w += sum.w;
      *p = o;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Up to 16 rows: ONE launch for 1..kMaxSegments ops over one vec, all three terms -- the grid of the batch-1 fused
// kernel (per op [CSR chunks | top-X slabs | pad to x8 | dense ranges]) with the split matrix-core role as its dense
// role.  (From 17 rows on the sparse terms are a launch of their own, as for the fp32 kernel: sqllm_sparse_batched.)
// The reference runs 1-3 dependent launches per op and one weight pass per batch row
// (squeezellm/quant_cuda_kernel.cu:580-657, :661-738).
// ------------------------------------------------------------------------------------------------
constexpr int kSmallRows = 16;

template <int BITS, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, 4)
sqllm_fused_small_split(const float* x, const GroupArgs ga, const float* xT, int Bp) {
  constexpr int T = WAVES * 64;
  // (the transposed-vec CSR role lays its LDS out for up to 64 rows: sqllm_roles.h, csr_role XTMODE)
  __shared__ __attribute__((aligned(16))) float lds[cmax(split_lds_floats(BITS, WAVES),
                                                         cmax(kCsrSpanMax + cmax(kCsrSpanMax, 64 * (kCsrXtSpan + 1) + 3 * kCsrChunk), kTopxLds))];
  // one round of scalar loads for the block table and segment 0 (see sqllm_fused_matvec)
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
    dense_role_mfma_split<BITS, 1, WAVES>(x, reinterpret_cast<const u32x4*>(sg.q), sg.y, sg.lut, gm.K, gm.N, gm.batch, 0, d, gm.col_tiles,
                                          gm.units_total, gm.units_per_wg,
                                          gm.dense_blocks == gm.col_tiles * gm.k_slices ? gm.k_slices * gm.units_per_wg : gm.units_total, lds);
  } else if (bid < gm.csr_blocks) {
    // vec transposed (xT[k][row], written by sqllm_transpose_vec just before this launch): ONE 64-byte read per non-zero
    // serves all the rows; without scratch, gathers from vec itself, one per non-zero and row
    if (xT) csr_role<T, 1, float, float, true>(x, sg.y, sg.rows, sg.cols, sg.vals, gm.nnz, gm.K, gm.N, 0, gm.batch, bid, lds, nullptr, 0, xT, Bp);
    else csr_role<T, kSmallRows, float, float>(x, sg.y, sg.rows, sg.cols, sg.vals, gm.nnz, gm.K, gm.N, 0, gm.batch, bid, lds, nullptr, 0);
  } else if (bid < gm.csr_blocks + gm.topx_blocks) {
    topx_role<T, float, float>(x, sg.y, sg.full_rows, sg.full_idx, gm.topX, gm.K, gm.N, 0, gm.batch, bid - gm.csr_blocks, lds);
  }
}

namespace {

template <int BITS, int MB>
hipError_t launch_split_inst(const LaunchArgs& a, hipStream_t stream) {
  const KernelGeom& gm = a.ga.seg[0].gm;
  dim3 grid(gm.dense_blocks, (gm.batch + 16 * MB - 1) / (16 * MB));
  auto kern = sqllm_fused_batched_split<BITS, MB, kWaves>;
  const float* x = static_cast<const float*>(a.x);
  if (a.ev_start || a.ev_stop) hipExtLaunchKernelGGL(kern, grid, dim3(kWaves * 64), 0, stream, a.ev_start, a.ev_stop, 0, x, a.ga);
  else hipLaunchKernelGGL(kern, grid, dim3(kWaves * 64), 0, stream, x, a.ga);
  return hipGetLastError();
}

template <int BITS>
hipError_t launch_split_bits(const LaunchArgs& a, hipStream_t stream) {
  switch (mfma_row_blocks(a.ga.seg[0].gm.batch)) {
    case 1: return launch_split_inst<BITS, 1>(a, stream);
    case 2: return launch_split_inst<BITS, 2>(a, stream);
    default: return launch_split_inst<BITS, 4>(a, stream);
  }
}

template <int BITS>
hipError_t launch_wide_bits(const LaunchArgs& a, hipStream_t stream) {
  const KernelGeom& gm = a.ga.seg[0].gm;
  dim3 grid(gm.dense_blocks);
  const int sliced_units = gm.k_slices > 1 ? (gm.dense_blocks - a.wide_full_units) / gm.k_slices : 0;
  float* slabs = sliced_units > 0 ? a.wide_slabs : nullptr;
  hipEvent_t stop = slabs ? nullptr : a.ev_stop;  // (with slabs the op ends with the reduce launch)
  if (a.planes) {
    auto kern = sqllm_fused_wide<BITS, true>;
    if (a.ev_start || stop) hipExtLaunchKernelGGL(kern, grid, dim3(kWaves * 64), 0, stream, a.ev_start, stop, 0, a.planes, a.plane_flags, a.wide_full_units, slabs, a.ga);
    else hipLaunchKernelGGL(kern, grid, dim3(kWaves * 64), 0, stream, a.planes, a.plane_flags, a.wide_full_units, slabs, a.ga);
  } else {
    auto kern = sqllm_fused_wide<BITS, false>;
    const uint32_t* none = nullptr;
    if (a.ev_start || stop) hipExtLaunchKernelGGL(kern, grid, dim3(kWaves * 64), 0, stream, a.ev_start, stop, 0, a.x, none, a.wide_full_units, slabs, a.ga);
    else hipLaunchKernelGGL(kern, grid, dim3(kWaves * 64), 0, stream, a.x, none, a.wide_full_units, slabs, a.ga);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || !slabs) return e;
  const Segment& sg = a.ga.seg[0];
  dim3 rgrid(sliced_units * kWideTiles);
  if (a.ev_stop) hipExtLaunchKernelGGL(sqllm_wide_reduce, rgrid, dim3(256), 0, stream, nullptr, a.ev_stop, 0, (const float*)slabs, sg.y, gm.N, gm.batch, gm.col_tiles, a.wide_full_units, gm.k_slices);
  else hipLaunchKernelGGL(sqllm_wide_reduce, rgrid, dim3(256), 0, stream, (const float*)slabs, sg.y, gm.N, gm.batch, gm.col_tiles, a.wide_full_units, gm.k_slices);
  return hipGetLastError();
}

}  // namespace

// 1..kMaxSegments ops over one vec (a.ga), up to kSmallRows rows: all three terms of every op in one launch
hipError_t launch_small_split(int bits, const LaunchArgs& a, hipStream_t stream) {
  if (a.ga.seg[0].gm.batch > kSmallRows) return hipErrorInvalidValue;
  dim3 grid(a.ga.block0[a.ga.n_seg]);
  const float* x = static_cast<const float*>(a.x);
  if (bits == 4) {
    auto kern = sqllm_fused_small_split<4, kWaves>;
    if (a.ev_start || a.ev_stop) hipExtLaunchKernelGGL(kern, grid, dim3(kWaves * 64), 0, stream, a.ev_start, a.ev_stop, 0, x, a.ga, a.xT, a.Bp);
    else hipLaunchKernelGGL(kern, grid, dim3(kWaves * 64), 0, stream, x, a.ga, a.xT, a.Bp);
  } else {
    auto kern = sqllm_fused_small_split<3, kWaves>;
    if (a.ev_start || a.ev_stop) hipExtLaunchKernelGGL(kern, grid, dim3(kWaves * 64), 0, stream, a.ev_start, a.ev_stop, 0, x, a.ga, a.xT, a.Bp);
    else hipLaunchKernelGGL(kern, grid, dim3(kWaves * 64), 0, stream, x, a.ga, a.xT, a.Bp);
  }
  return hipGetLastError();
}

// vec [batch, K] -> bf16 planes in fragment order + lo flags (sqllm_split_vec); `planes` holds split_planes_chunks(batch, K) chunks
hipError_t split_vec(const float* x, void* planes, uint32_t* flags, int batch, int K, hipStream_t stream, hipEvent_t ev_start) {
  u32x4* out = static_cast<u32x4*>(planes);
  const uint32_t n_frag = (uint32_t)(split_planes_chunks(batch, K) / 3);  // (rb, kb, lane) triples
  if (ev_start) hipExtLaunchKernelGGL(sqllm_split_vec, dim3(kSplitFlagWgs), dim3(256), 0, stream, ev_start, nullptr, 0, x, out, n_frag, flags, batch, K);
  else hipLaunchKernelGGL(sqllm_split_vec, dim3(kSplitFlagWgs), dim3(256), 0, stream, x, out, n_frag, flags, batch, K);
  return hipGetLastError();
}

// one op (a.ga.seg[0]), operator ABI, batch rows through the bf16 matrix cores with split operands (dense term only);
// a.wide: the wide form, A operands ready-made from split_vec's planes (a.planes) or split in registers
hipError_t launch_batched_mfma_split(int bits, const LaunchArgs& a, hipStream_t stream) {
  if (a.wide) return bits == 4 ? launch_wide_bits<4>(a, stream) : launch_wide_bits<3>(a, stream);
  return bits == 4 ? launch_split_bits<4>(a, stream) : launch_split_bits<3>(a, stream);
}

}  // namespace sqllm
