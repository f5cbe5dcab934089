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
// The WIDE form -- workgroups of eight column tiles, vec split once into bf16 planes -- is sqllm_mfma_wide.hip.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <type_traits>

#include "sqllm_kernels.h"

#include "sqllm_decode.h"
#include "sqllm_roles.h"
#include "sqllm_split_common.h"
#include "sqllm_probe.h"

namespace sqllm {

namespace {

// FOLD: the op's CSR term is walked by these workgroups themselves (csr_tile_fold_staged, sqllm_roles.h; MB == 1 only):
// csr_rows / csr_cols / csr_vals are the op's CSR (csr_rows null: no such term).  CSR rows are output channels, so the
// non-zeros of the piece's 64-column tile are one contiguous range of cols / vals; the pieces of a tile cut it in proportion
// to their K ranges.  ALL waves take part: the first columns / values of the piece's share and their row searches go out
// before the dense loop and are STAGED in the slab area of LDS (idle until the epilogue), the walk itself runs after the
// loop -- lane = (group, batch row), a group of 8 / 16 lanes takes a run of the share one non-zero per step, groups dealt
// round-robin (static: no ticket) -- and its sums meet the dense partial sums in the tile's own epilogue.  (One wave
// walking beside seven decoding, with groups by LDS ticket, was measured and dropped: profiles/r05_walker_wave.txt.)
// XMODE 0: vec as fp32 rows, split in registers; 3: vec ALREADY SPLIT into three bf16 planes in fragment order (`planes`:
// sqllm_prepare_small, row block 0 -- MB == 1 only), the A fragments are loaded ready-made.
template <int BITS, int MB, int WAVES, bool FOLD = false, int XMODE = 0>
__device__ __forceinline__ void dense_role_mfma_split(const float* __restrict__ x, const u32x4* __restrict__ q,
                                                      float* __restrict__ y, const float* __restrict__ lut, int K, int N,
                                                      int batch, int m0, int bid, int n_col_tiles, int units_total,
                                                      int units_per_wg, int units_stride, float* lds,
                                                      const int* __restrict__ csr_rows = nullptr, const int* __restrict__ csr_cols = nullptr,
                                                      const float* __restrict__ csr_vals = nullptr, const float* __restrict__ xT = nullptr,
                                                      unsigned long long* tl = nullptr, const char* __restrict__ planes = nullptr) {
  static_assert(!FOLD || MB == 1, "the folded CSR walk serves one block of 16 rows");
  static_assert(XMODE == 0 || MB == 1, "planes of row block 0 only");
  static_assert(!FOLD || WAVES * 16 * 64 >= 2 * kFoldStage, "the walk stages columns and values in the slab area");
  constexpr int NX = XMODE == 0 ? 2 * MB : XMODE * MB;  // 16-byte registers of one phase's vec values
  using F = Fmt<BITS>;
  constexpr int L = F::kLut, R = F::kRows, KU = F::kK;
  constexpr int NPH = KU / 8;  // phases of 8 k's per unit (4-bit: 1, 3-bit: 4)
  constexpr int T = WAVES * 64;
  __builtin_amdgcn_s_waitcnt(0);  // clean slate for the compiler's wait-count model (see dense_role)
  const int tid = threadIdx.x;
  SQLLM_PROBE_ENTRY(tl, tid == 0);  // entry (+ where: XCC, CU)
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, grp = lane >> 4;
  constexpr int kCbBytes = split_codebook_bytes(BITS);
  float* slabs = lds + kCbBytes / 4;
  float* ssum = slabs + WAVES * 16 * 64;                   // FOLD: [16][64] sums of the CSR walk (zero between pieces)
  int* srp = reinterpret_cast<int*>(ssum + kFoldSum);      // FOLD: the tile's row pointers
  const bool fold = FOLD && csr_rows != nullptr;
  if constexpr (FOLD) {
    for (int i = tid; i < kFoldSum; i += T) ssum[i] = 0.f;  // (visible after the first piece's staging barrier)
  }
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
  int rpv = 0;
  if constexpr (FOLD) {  // row pointer (tid % 128) of the tile, clamped at N: an unconditional load beside the codebook's
    int rc = col0 + (tid & (kFoldRp - 1));
    if (rc > N) rc = N;
    rpv = (fold ? csr_rows : reinterpret_cast<const int*>(lut))[fold ? rc : 0];
  }
  const int n_groups_wg = (u_end - u_beg + 3) / 4;
  const int n_g = n_groups_wg > wave ? (n_groups_wg - wave + WAVES - 1) / WAVES : 0;
  int cidx = col0 / 4 + i16;
  if (cidx > row_stride - 1) cidx = row_stride - 1;
  const uint32_t lane_bytes = 16u * (uint32_t)cidx;
  auto group_unit = [&](int t) {  // unit of this lane's row in group t of the piece (may be >= u_end)
    return u_beg + 4 * t + grp;
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
      // fragment order (see dense_role_mfma_wide): 4-bit -- the group's 32 k's are ONE k block, lane for lane; 3-bit -- a
      // lane row's unit is a k block of its own, phase ph = its quarter; past the K range: the zero k block
      const uint32_t kb = (gu < u_end && gu >= u_beg) ? (BITS == 4 ? (uint32_t)u >> 2 : (uint32_t)u) : (uint32_t)K / 32u;
      const uint32_t lp = BITS == 4 ? (uint32_t)lane : (uint32_t)(16 * ph + i16);
      const uint32_t off = 3072u * kb + 16u * lp;
#pragma unroll
      for (int pl = 0; pl < XMODE; ++pl) dx[pl] = *reinterpret_cast<const u32x4*>(planes + off + 1024 * pl);
    }
  };
  u32x4 wa[R], wb[R];
  u32x4 xa[NX], xb[NX];
  load_w(wave, wa);
  load_x(wave, 0, xa);
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  // ---- stage the codebooks, split ----
  {
    char* base = reinterpret_cast<char*>(lds);
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      *reinterpret_cast<u32x2*>(base + 8 * (tid + T * i)) = split_entry(ev[i]);
    }
    if constexpr (FOLD) srp[tid & (kFoldRp - 1)] = rpv;
  }
  f32x4 acc[MB][4];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[mb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();  // codebooks staged (and everybody has left the previous piece's slabs)
  SQLLM_PROBE(tl, 1, tid == 0);  // codebooks staged
  // FOLD: this piece's share of the tile's non-zeros, and the first kFoldPre columns / values per thread on their way
  constexpr int kFoldPre = 3;
  int sp_beg = 0, sp_end = 0;
  int pre_c[kFoldPre];
  float pre_v[kFoldPre];
  unsigned pre_r = 0;
  if constexpr (FOLD) {
    if (fold) {
      fold_piece_share(srp, u_beg, u_end, units_total, &sp_beg, &sp_end);
      sp_beg = __builtin_amdgcn_readfirstlane(sp_beg);
      sp_end = __builtin_amdgcn_readfirstlane(sp_end);
    }
    {  // their CSR rows (local to the tile), six bits each
      int ea[kFoldPre], rr[kFoldPre];
#pragma unroll
      for (int i = 0; i < kFoldPre; ++i) ea[i] = sp_beg + tid + T * i;
      fold_rows_of<kFoldPre>(srp, ea, rr);
#pragma unroll
      for (int i = 0; i < kFoldPre; ++i) pre_r |= (unsigned)rr[i] << (6 * i);
    }
#pragma unroll
    for (int i = 0; i < kFoldPre; ++i) {
      int ee = sp_beg + tid + T * i;
      if (ee > sp_end - 1) ee = sp_end - 1;
      if (ee < 0) ee = 0;
      // (unconditional loads -- see the codebook's; without the term: the codebook pointer)
      pre_c[i] = (fold ? csr_cols : reinterpret_cast<const int*>(lut))[fold ? ee : 0];
      pre_v[i] = (fold ? csr_vals : lut)[fold ? ee : 0];
    }
  }

  auto phase = [&](const u32x4 (&t)[R], auto ph_tag, const u32x4 (&dx)[NX], int g) {
    split_phase<BITS, MB, XMODE, decltype(ph_tag)::value>(t, dx, group_unit(g) < u_end, lane_off, 0u, acc);
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using P2 = std::integral_constant<int, 2>;
  using P3 = std::integral_constant<int, 3>;
  // decode group g out of (w, xcur = its phase-0 vec values); the wave's NEXT group gn: its weights (into wn) are loaded
  // before the first phase, its phase-0 values (into xn) before the last.  With four phases the values of phase p + 1 are
  // loaded while phase p runs, alternating between xcur and xo -- the next group's phase 0 lands in xcur again (xn == xcur).
  auto decode_group = [&](int g, int gn, const u32x4 (&w)[R], u32x4 (&xcur)[NX], u32x4 (&wn)[R], u32x4 (&xn)[NX], u32x4 (&xo)[NX]) {
    load_w(gn, wn);
    if constexpr (NPH == 1) {
      load_x(gn, 0, xn);
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
      load_x(gn, 0, xn);
      __builtin_amdgcn_sched_barrier(0);
      phase(w, P3{}, xo, g);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // (Measured and dropped twice, profiles/r05_walker_wave.txt and r05_tickets.txt: the groups handed out by an LDS ticket
  // instead of round-robin -- the waves of a workgroup leave this loop up to 2.8 us apart -- 2-4 % slower.)
  for (int g = 0; g < n_g; g += 2) {  // the wave's groups: wave, wave + WAVES, ...
    const int t = wave + WAVES * g;
    if constexpr (NPH == 1) {
      decode_group(t, t + WAVES, wa, xa, wb, xb, xb);
      decode_group(t + WAVES, t + 2 * WAVES, wb, xb, wa, xa, xa);
    } else {
      decode_group(t, t + WAVES, wa, xa, wb, xa, xb);
      decode_group(t + WAVES, t + 2 * WAVES, wb, xa, wa, xa, xb);
    }
  }
  SQLLM_PROBE(tl, 2, tid == 0);  // wave 0 done decoding
  SQLLM_PROBE(tl, 7, tid == T - 64);  // the last wave done decoding
  if constexpr (FOLD) {
    if (fold)  // (walked BEFORE the loop by every other CU-load of workgroups, so that neighbours on a CU alternate: no gain -- profiles/r05_walk_first.txt)
      csr_tile_fold_staged<T, 20, kFoldPre>(x, xT, csr_cols, csr_vals, K, m0, batch - m0 < 16 ? batch - m0 : 16, sp_beg, sp_end, srp, ssum,
                                           reinterpret_cast<int*>(slabs), tid, pre_c, pre_v, pre_r, tl);
  }
  SQLLM_PROBE(tl, 5, tid == 0);  // CSR share walked (all waves)
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
      if (!is_finite_f32(sum) && r < batch && col < N)  // non-finite operands: the reference's fp32 chain (sqllm_split_common.h)
        sum = dense_term_fp32<BITS>(x + (size_t)r * K, reinterpret_cast<const uint32_t*>(q), lut, N, col, u_beg * KU, u_end * KU);
      if constexpr (FOLD) {  // the CSR walk's sums; zero again for the next piece (this thread is the element's only reader)
        const int se = (e >> 6) * kFoldSumStride + (e & 63);
        sum += ssum[se];
        ssum[se] = 0.f;
      }
      if (r < batch && col < N) acc_add(y + (size_t)r * N + col, sum);
    }
  }
  SQLLM_PROBE(tl, 6, tid == 0);  // atomics issued
  }  // pieces
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

// The same with the op's sparse terms in the grid (17 rows up to the wide form's switch-over, one op per launch):
// blockIdx.x = [CSR chunks | top-X slabs | pad to x8 | dense ranges], blockIdx.y = passes of 16 * MB rows.  As launches
// of their own the sparse terms are pure latency on an underfilled chip -- 331 workgroups for a 13B gate/up op, 25 / 58 us
// at 32 / 64 rows against 39 / 58 for the dense kernel (profiles/r04_kt_mid_rows.txt) -- here they run beside the
// dense workgroups.  xT: the transposed copy of vec (sqllm_transpose_vec, launched before), or null: gathers.
template <int BITS, int MB, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, (MB == 1 || (BITS == 4 && MB == 2)) ? 4 : 2)
sqllm_fused_batched_split_all(const float* x, const GroupArgs ga, const float* xT, int Bp) {
  constexpr int T = WAVES * 64;
  __shared__ __attribute__((aligned(16))) float lds[cmax(split_lds_floats(BITS, WAVES),
                                                         cmax(kCsrSpanMax + cmax(kCsrSpanMax, 64 * (kCsrXtSpan + 1) + 3 * kCsrChunk), kTopxLds))];
  const Segment sg = ga.seg[0];
  asm volatile("" ::SQLLM_SEG_OPERANDS(sg), "s"(x), "s"(xT));
  __builtin_amdgcn_sched_barrier(0);
  const KernelGeom& gm = sg.gm;
  const int m0 = blockIdx.y * 16 * MB;
  int rows_here = gm.batch - m0;
  if (rows_here > 16 * MB) rows_here = 16 * MB;
  const int bid = blockIdx.x;
  const int d = bid - gm.dense_block0;
  if (d >= 0 && d < gm.dense_blocks) {
    dense_role_mfma_split<BITS, MB, WAVES>(x, reinterpret_cast<const u32x4*>(sg.q), sg.y, sg.lut, gm.K, gm.N, gm.batch, m0, d, gm.col_tiles,
                                           gm.units_total, gm.units_per_wg,
                                           gm.dense_blocks == gm.col_tiles * gm.k_slices ? gm.k_slices * gm.units_per_wg : gm.units_total, lds);
  } else if (bid < gm.csr_blocks) {
    if (xT) {
      csr_role<T, 1, float, float, true>(x, sg.y, sg.rows, sg.cols, sg.vals, gm.nnz, gm.K, gm.N, m0, rows_here, bid, lds, nullptr, 0, xT, Bp);
    } else {
      constexpr int CBT = 32;  // (see sqllm_sparse_batched)
      for (int bb = 0; bb < rows_here; bb += CBT) {
        if (bb) __syncthreads();
        csr_role<T, CBT, float, float>(x, sg.y, sg.rows, sg.cols, sg.vals, gm.nnz, gm.K, gm.N, m0 + bb, rows_here - bb < CBT ? rows_here - bb : CBT, bid, lds,
                                       nullptr, 0);
      }
    }
  } else if (bid < gm.csr_blocks + gm.topx_blocks) {
    topx_role<T, float, float, false, NoGate, 4>(x, sg.y, sg.full_rows, sg.full_idx, gm.topX, gm.K, gm.N, m0, rows_here, bid - gm.csr_blocks, lds);  // (passes of 4 rows)
  }
}

// ------------------------------------------------------------------------------------------------
// Up to 16 rows: ONE launch for 1..kMaxSegments ops over one vec, all three terms -- per op [top-X slabs | pad to x8 |
// dense ranges], the split matrix-core role as the dense role, and the CSR term FOLDED into it (csr_tile_fold,
// sqllm_roles.h: every dense workgroup walks the non-zeros of its own 64 output channels; gm.fold_csr, gm.csr_blocks == 0).
// Until round 5 the CSR term was a role of its own here (one workgroup per 1024 non-zeros, first in the grid): with its
// LDS (68 KB) and registers (106) the kernel held two workgroups per CU, the chunk workgroups took the whole first round
// of slots for their chain of dependent round trips, and a 13B s45 decoder layer paid 50 us for them at 8 rows, 80 at 16
// (profiles/r04_small_batch_layer.txt).  Now: 53 KB and <= 80 registers at 4 bits, three workgroups per CU.
// (From 17 rows on the sparse terms are workgroups in the grid or a launch of their own: sqllm_fused_batched_split_all,
// sqllm_sparse_batched.)  The reference runs 1-3 dependent launches per op and one weight pass per batch row
// (squeezellm/quant_cuda_kernel.cu:580-657, :661-738).
// ------------------------------------------------------------------------------------------------
constexpr int kSmallRows = 16;

// XM 3: vec comes already split into bf16 planes (`planes`, written with xT by sqllm_prepare_small); 0: fp32 rows.
template <int BITS, int WAVES, int XM>
__global__ void __launch_bounds__(WAVES * 64, 4)
sqllm_fused_small_split(const float* x, const GroupArgs ga, const float* xT, const char* planes) {
  constexpr int T = WAVES * 64;
  __shared__ __attribute__((aligned(16))) float lds[cmax(split_lds_floats(BITS, WAVES) + kFoldSum + kFoldRp, kTopxLds)];
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
    dense_role_mfma_split<BITS, 1, WAVES, true, XM>(x, reinterpret_cast<const u32x4*>(sg.q), sg.y, sg.lut, gm.K, gm.N, gm.batch, 0, d, gm.col_tiles,
                                                gm.units_total, gm.units_per_wg,
                                                gm.dense_blocks == gm.col_tiles * gm.k_slices ? gm.k_slices * gm.units_per_wg : gm.units_total, lds,
                                                gm.nnz > 0 ? sg.rows : nullptr, sg.cols, sg.vals, xT, SQLLM_PROBE_PTR(sg), planes);
  } else if (bid < gm.topx_blocks) {
    // top-X slabs: gm.topx_blocks workgroups share the op's ceil(K / kTopxRows) slabs (with the transposed vec: 8-16
    // workgroups of several slabs each, all batch rows at once; without: one slab each, passes of 8 rows)
    const int slabs = (gm.K + kTopxRows - 1) / kTopxRows;
    const int spw = (slabs + gm.topx_blocks - 1) / gm.topx_blocks;
    const int s0 = bid * spw, s1 = s0 + spw < slabs ? s0 + spw : slabs;
    if (xT && gm.topX <= 16) {
      const int lr = gm.batch <= 2 ? 1 : gm.batch <= 4 ? 2 : gm.batch <= 8 ? 3 : 4;
      if (s0 < s1) topx_role_xt<T>(xT, lr, sg.y, sg.full_rows, sg.full_idx, gm.topX, gm.K, gm.N, gm.batch, s0, s1, lds);
    } else {
      for (int sl = s0; sl < s1; ++sl) {
        if (sl > s0) __syncthreads();
        topx_role<T, float, float, false, NoGate, 8>(x, sg.y, sg.full_rows, sg.full_idx, gm.topX, gm.K, gm.N, 0, gm.batch, sl, lds);
      }
    }
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

template <int BITS, int MB>
hipError_t launch_split_all_inst(const LaunchArgs& a, hipStream_t stream) {
  const KernelGeom& gm = a.ga.seg[0].gm;
  dim3 grid(gm.dense_block0 + gm.dense_blocks, (gm.batch + 16 * MB - 1) / (16 * MB));
  auto kern = sqllm_fused_batched_split_all<BITS, MB, kWaves>;
  const float* x = static_cast<const float*>(a.x);
  if (a.ev_start || a.ev_stop) hipExtLaunchKernelGGL(kern, grid, dim3(kWaves * 64), 0, stream, a.ev_start, a.ev_stop, 0, x, a.ga, a.xT, a.Bp);
  else hipLaunchKernelGGL(kern, grid, dim3(kWaves * 64), 0, stream, x, a.ga, a.xT, a.Bp);
  return hipGetLastError();
}

template <int BITS>
hipError_t launch_split_all_bits(const LaunchArgs& a, hipStream_t stream) {
  switch (a.row_blocks > 0 ? a.row_blocks : mfma_row_blocks(a.ga.seg[0].gm.batch)) {
    case 1: return launch_split_all_inst<BITS, 1>(a, stream);
    case 2: return launch_split_all_inst<BITS, 2>(a, stream);
    default: return launch_split_all_inst<BITS, 4>(a, stream);
  }
}

template <int BITS>
hipError_t launch_split_bits(const LaunchArgs& a, hipStream_t stream) {
  switch (mfma_row_blocks(a.ga.seg[0].gm.batch)) {
    case 1: return launch_split_inst<BITS, 1>(a, stream);
    case 2: return launch_split_inst<BITS, 2>(a, stream);
    default: return launch_split_inst<BITS, 4>(a, stream);
  }
}

}  // namespace

// 1..kMaxSegments ops over one vec (a.ga), up to kSmallRows rows: all three terms of every op in one launch
namespace {
template <int BITS, int XM>
hipError_t launch_small_split_inst(const LaunchArgs& a, hipStream_t stream) {
  dim3 grid(a.ga.block0[a.ga.n_seg]);
  const float* x = static_cast<const float*>(a.x);
  const char* planes = static_cast<const char*>(a.planes);
  auto kern = sqllm_fused_small_split<BITS, kWaves, XM>;
  if (a.ev_start || a.ev_stop) hipExtLaunchKernelGGL(kern, grid, dim3(kWaves * 64), 0, stream, a.ev_start, a.ev_stop, 0, x, a.ga, a.xT, planes);
  else hipLaunchKernelGGL(kern, grid, dim3(kWaves * 64), 0, stream, x, a.ga, a.xT, planes);
  return hipGetLastError();
}
}  // namespace

// (a.planes: vec as bf16 planes of row block 0, or null: fp32 rows split in registers)
hipError_t launch_small_split(int bits, const LaunchArgs& a, hipStream_t stream) {
  if (a.ga.seg[0].gm.batch > kSmallRows) return hipErrorInvalidValue;
  if (bits == 4) return a.planes ? launch_small_split_inst<4, 3>(a, stream) : launch_small_split_inst<4, 0>(a, stream);
  return a.planes ? launch_small_split_inst<3, 3>(a, stream) : launch_small_split_inst<3, 0>(a, stream);
}

// one op (a.ga.seg[0]) in the tile form with its CSR / top-X workgroups in the same grid (a.xT: transposed vec or null)
hipError_t launch_batched_mfma_split_all(int bits, const LaunchArgs& a, hipStream_t stream) {
  return bits == 4 ? launch_split_all_bits<4>(a, stream) : launch_split_all_bits<3>(a, stream);
}

// one op (a.ga.seg[0]), operator ABI, batch rows through the bf16 matrix cores with split operands (dense term only);
// a.wide: the wide form (sqllm_mfma_wide.hip)
hipError_t launch_batched_mfma_split(int bits, const LaunchArgs& a, hipStream_t stream) {
  if (a.wide) return launch_batched_mfma_wide(bits, a, stream);
  return bits == 4 ? launch_split_bits<4>(a, stream) : launch_split_bits<3>(a, stream);
}

}  // namespace sqllm
