// sqllm_mfma_wide.hip -- the WIDE form of the split matrix-core kernel (the arithmetic: sqllm_mfma_split.hip; shared device
// code: sqllm_split_common.h): the dense term of the *_batched operators once the op is large enough (sqllm_capi.hip:
// takes_wide_path; 13B shapes: from 128-256 rows).  Reference: one weight pass PER BATCH ROW,
// squeezellm/quant_cuda_kernel.cu:884-979 / :982-1038.
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
#include "sqllm_split_common.h"

namespace sqllm {

namespace {

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
          // (pinned to its slot: the scheduling barriers bind the machine scheduler only -- the packing of a phase's LAST
          // column, whose results leave the phase, was sunk behind all its matrix instructions by an earlier pass:
          // tests/test_codegen_cpu.py found 19 of them back to back)
          asm volatile("" : "+v"(Bn[4 * kind + i]));
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
// adds its 64 x 64 results straight to mul (or, as a K slice, leaves them as a slab for sqllm_wide_reduce).  Grid and
// work units: see sqllm_fused_wide.
// ------------------------------------------------------------------------------------------------
constexpr int kWideTiles = 8;  // column tiles per workgroup = waves
constexpr int kWideSlabFloats = 64 * kTileN;  // a wave's 64 x 64 sums

template <int BITS, int XMODE>
__device__ __forceinline__ void dense_role_mfma_wide(const void* __restrict__ xv, const u32x4* __restrict__ q,
                                                     float* __restrict__ y, const float* __restrict__ lut, int K, int N, int batch,
                                                     int m0, int ct, int u_beg, int u_end, bool atomic, float* __restrict__ slab,
                                                     const float* __restrict__ x32) {
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
  // non-finite operands (sqllm_split_common.h): a non-finite sum is recomputed as the reference's fp32 chain over this
  // workgroup's k's, out of the fp32 vec (x32) -- cold code
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (!is_finite_f32(acc[mb][j][e])) {
          const int r = m0 + 16 * mb + 4 * grp + e, col = c0 + j;
          if (r < batch && col < N)
            acc[mb][j][e] = dense_term_fp32<BITS>(x32 + (size_t)r * K, reinterpret_cast<const uint32_t*>(q), lut, N, col, u_beg * KU, u_end * KU);
        }
      }
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
sqllm_fused_wide(const void* xv, const uint32_t* flags, int full_units, float* slabs, const GroupArgs ga, const float* x32) {
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
    if (has_lo) dense_role_mfma_wide<BITS, 3>(xv, q, sg.y, sg.lut, gm.K, gm.N, gm.batch, m0, ct, u_beg, u_end, sliced, slab, x32);
    else dense_role_mfma_wide<BITS, 2>(xv, q, sg.y, sg.lut, gm.K, gm.N, gm.batch, m0, ct, u_beg, u_end, sliced, slab, x32);
  } else {
    dense_role_mfma_wide<BITS, 0>(xv, q, sg.y, sg.lut, gm.K, gm.N, gm.batch, m0, ct, u_beg, u_end, sliced, slab, x32);
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

namespace {

template <int BITS>
hipError_t launch_wide_bits(const LaunchArgs& a, hipStream_t stream) {
  const KernelGeom& gm = a.ga.seg[0].gm;
  dim3 grid(gm.dense_blocks);
  const int sliced_units = gm.k_slices > 1 ? (gm.dense_blocks - a.wide_full_units) / gm.k_slices : 0;
  float* slabs = sliced_units > 0 ? a.wide_slabs : nullptr;
  hipEvent_t stop = slabs ? nullptr : a.ev_stop;  // (with slabs the op ends with the reduce launch)
  if (a.planes) {
    auto kern = sqllm_fused_wide<BITS, true>;
    if (a.ev_start || stop) hipExtLaunchKernelGGL(kern, grid, dim3(kWaves * 64), 0, stream, a.ev_start, stop, 0, a.planes, a.plane_flags, a.wide_full_units, slabs, a.ga, static_cast<const float*>(a.x));
    else hipLaunchKernelGGL(kern, grid, dim3(kWaves * 64), 0, stream, a.planes, a.plane_flags, a.wide_full_units, slabs, a.ga, static_cast<const float*>(a.x));
  } else {
    auto kern = sqllm_fused_wide<BITS, false>;
    const uint32_t* none = nullptr;
    if (a.ev_start || stop) hipExtLaunchKernelGGL(kern, grid, dim3(kWaves * 64), 0, stream, a.ev_start, stop, 0, a.x, none, a.wide_full_units, slabs, a.ga, static_cast<const float*>(a.x));
    else hipLaunchKernelGGL(kern, grid, dim3(kWaves * 64), 0, stream, a.x, none, a.wide_full_units, slabs, a.ga, static_cast<const float*>(a.x));
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

// vec [batch, K] -> bf16 planes in fragment order + lo flags (sqllm_split_vec); `planes` holds split_planes_chunks(batch, K) chunks
hipError_t split_vec(const float* x, void* planes, uint32_t* flags, int batch, int K, hipStream_t stream, hipEvent_t ev_start) {
  u32x4* out = static_cast<u32x4*>(planes);
  const uint32_t n_frag = (uint32_t)(split_planes_chunks(batch, K) / 3);  // (rb, kb, lane) triples
  if (ev_start) hipExtLaunchKernelGGL(sqllm_split_vec, dim3(kSplitFlagWgs), dim3(256), 0, stream, ev_start, nullptr, 0, x, out, n_frag, flags, batch, K);
  else hipLaunchKernelGGL(sqllm_split_vec, dim3(kSplitFlagWgs), dim3(256), 0, stream, x, out, n_frag, flags, batch, K);
  return hipGetLastError();
}

// one op (a.ga.seg[0]) in the wide form: a.planes = vec as split_vec's planes (or null: split in registers), a.wide_slabs =
// room for the K slices' sums (or null: they add atomically)
hipError_t launch_batched_mfma_wide(int bits, const LaunchArgs& a, hipStream_t stream) {
  return bits == 4 ? launch_wide_bits<4>(a, stream) : launch_wide_bits<3>(a, stream);
}

}  // namespace sqllm
