// sqllm_kernels.hip -- the product's kernel instantiations and launchers: the fused batch-tile kernel (template in
// sqllm_fused.h), the wide-batch matrix-core kernel, the small-batch column-lane kernel, the wide-batch sparse
// launch, the vec transpose and the CSR check.
#include "sqllm_fused.h"
#include "sqllm_split_common.h"

namespace sqllm {

// measurement library only (csrc/experimental/sqllm_ablation.hip installs it; null in the product): ablation
// instantiations of the fused kernel and the calibration kernels, by LaunchArgs::ablate
bool (*g_fused_variant)(int bits, const LaunchArgs& a, hipStream_t stream, hipError_t* err) = nullptr;

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
                                                int units_per_wg, int units_stride, float* lds) {
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
  // (a tile is units_stride long in the flattened space: units_total, or -- tile-aligned ranges, see dense_role_cols --
  // a whole number of ranges)
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
                                   (int)blockIdx.x, gm.col_tiles, gm.units_total, gm.units_per_wg,
                                   gm.dense_blocks == gm.col_tiles * gm.k_slices ? gm.k_slices * gm.units_per_wg : gm.units_total, lds);
}

// The sparse terms of a wide-batch op, as a launch of their own: inside the matrix-core kernel the
// CSR workgroups would inherit its register allocation (one or two workgroups per CU) and run their
// latency-bound loops without anybody to hide behind -- measured 3.3 ms of a 5.7 ms launch at
// 2048 rows.  Blocks of kSparsePassRows = 128 rows (blockIdx.y); blockIdx.x = CSR chunks, then top-X slabs.
//   xT != null: the CSR role reads the transposed copy of vec (lane = batch row);
//   xT == null (no scratch, or the stream is capturing): it gathers from vec, 32 rows at a time.
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
sqllm_sparse_batched(const float* x, const GroupArgs ga, const float* xT, int Bp, int pass_rows) {
  constexpr int T = WAVES * 64;
  __shared__ __attribute__((aligned(16))) float lds[cmax(kCsrSpanMax + cmax(kCsrSpanMax, 64 * (kCsrXtSpan + 1) + 3 * kCsrChunk), kTopxLds)];
  const Segment sg = ga.seg[0];  // the whole descriptor in one round of scalar loads (see sqllm_fused_matvec)
  asm volatile("" ::SQLLM_SEG_OPERANDS(sg), "s"(x));
  __builtin_amdgcn_sched_barrier(0);
  const KernelGeom& gm = sg.gm;
  const int sp = blockIdx.x;
  // blockIdx.y = a block of pass_rows rows (64, or kSparsePassRows = 128 where the grid stays large: launch_batched_sparse):
  // the CSR role with a transposed vec takes 128 two per lane where it can (csr_role), everything else in passes of 64 / 32
  const int m0 = blockIdx.y * pass_rows;
  int rows_here = gm.batch - m0;
  if (rows_here > pass_rows) rows_here = pass_rows;
  if (sp < gm.csr_blocks) {
    if (xT) {
      csr_role<T, 1, float, float, true>(x, sg.y, sg.rows, sg.cols, sg.vals, gm.nnz, gm.K, gm.N, m0, rows_here, sp, lds, nullptr, 0, xT, Bp, SQLLM_PROBE_PTR(sg));
    } else {
      constexpr int CBT = 32;  // every group of rows costs the chunk a zero / gather / flush round with its barriers
      for (int bb = 0; bb < rows_here; bb += CBT) {
        if (bb) __syncthreads();
        csr_role<T, CBT, float, float>(x, sg.y, sg.rows, sg.cols, sg.vals, gm.nnz, gm.K, gm.N, m0 + bb,
                                       rows_here - bb < CBT ? rows_here - bb : CBT, sp, lds, nullptr, 0);
      }
    }
  } else if (sp < gm.csr_blocks + gm.topx_blocks) {
    for (int bb = 0; bb < rows_here; bb += 64) {
      if (bb) __syncthreads();
      topx_role<T, float, float, false, NoGate, 8>(x, sg.y, sg.full_rows, sg.full_idx, gm.topX, gm.K, gm.N, m0 + bb, rows_here - bb < 64 ? rows_here - bb : 64,
                                 sp - gm.csr_blocks, lds);
    }
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
                                                int units_per_wg, int units_stride, float* lds) {
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

  // The ranges live in a flattened (column tile, unit) space in which a tile is units_stride long: = units_total
  // (contiguous ranges, some crossing a tile boundary: two pieces, two table builds) or, where the plan could cut
  // every tile into a whole number of ranges, that number x units_per_wg (>= units_total: no range crosses, the
  // last one of a tile is short).
  const unsigned total = (unsigned)n_col_tiles * (unsigned)units_stride;
  unsigned gpos = (unsigned)bid * (unsigned)units_per_wg;
  unsigned gend = gpos + (unsigned)units_per_wg;
  if (gend > total) gend = total;
  bool first = true;
  while (gpos < gend) {
    const int ct = __builtin_amdgcn_readfirstlane((int)(gpos / (unsigned)units_stride));  // (the division runs on the VALU)
    const int u0 = (int)(gpos - (unsigned)ct * (unsigned)units_stride);
    if (u0 >= units_total) { gpos = (unsigned)(ct + 1) * (unsigned)units_stride; continue; }  // (padding behind a tile's last range)
    int u1 = units_total;
    if ((unsigned)(u1 - u0) > gend - gpos) u1 = u0 + (int)(gend - gpos);
    gpos += (unsigned)(u1 - u0);
    if (u1 == units_total) gpos = (unsigned)(ct + 1) * (unsigned)units_stride;  // skip the padding
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
  // up to kMaxSegments ops over the same vec (q/k/v, gate/up): workgroup ids [block0[s], block0[s+1]) belong to op s.
  // Segment 0's descriptor and the block table in one round of scalar loads (see sqllm_fused_matvec), another
  // segment's in one more.
  Segment sg = ga.seg[0];
  const int n_seg = ga.n_seg, blk1 = ga.block0[1], blk2 = ga.block0[2], blk3 = ga.block0[3];
  asm volatile("" ::SQLLM_SEG_OPERANDS(sg), "s"(x), "s"(n_seg), "s"(blk1), "s"(blk2), "s"(blk3));
  __builtin_amdgcn_sched_barrier(0);
  int bid = blockIdx.x;
  int s = 0;
  if (n_seg > 1 && bid >= blk1) s = 1;
  if (n_seg > 2 && bid >= blk2) s = 2;
  if (n_seg > 3 && bid >= blk3) s = 3;
  if (s != 0) {
    sg = ga.seg[s];
    asm volatile("" ::SQLLM_SEG_OPERANDS(sg));
    bid -= s == 1 ? blk1 : s == 2 ? blk2 : blk3;
  }
  const KernelGeom& gm = sg.gm;
  const int b0 = blockIdx.y * BT;
  int nb = gm.batch - b0;
  if (nb > BT) nb = BT;
  const int d = bid - gm.dense_block0;
  const int sp = bid < gm.dense_block0 ? bid : -1;
  if (d >= 0 && d < gm.dense_blocks) {
    dense_role_cols<BITS, BT, WAVES>(x, sg.q, sg.y, sg.lut, gm.K, gm.N, b0, nb, d, gm.col_tiles, gm.units_total,
                                     gm.units_per_wg,
                                     gm.dense_blocks == gm.col_tiles * gm.k_slices ? gm.k_slices * gm.units_per_wg : gm.units_total, lds);
  } else if (sp >= 0 && sp < gm.csr_blocks) {
    csr_role<T, BT, float, float>(x, sg.y, sg.rows, sg.cols, sg.vals, gm.nnz, gm.K, gm.N, b0, nb, sp, lds, nullptr, 0);
  } else if (sp >= gm.csr_blocks && sp < gm.csr_blocks + gm.topx_blocks) {
    topx_role<T, float, float, false, NoGate, (BT < 4 ? BT : 4)>(x, sg.y, sg.full_rows, sg.full_idx, gm.topX, gm.K, gm.N, b0, nb, sp - gm.csr_blocks, lds);  // (passes of up to 4 rows: the kernel's 80 registers)
  }
}

template <int BITS, bool LIN>
static hipError_t launch_bt(const LaunchArgs& a, hipStream_t stream) {
  const int batch = a.ga.seg[0].gm.batch;
  if constexpr (!LIN) {
    switch (batch_tile_op(batch)) {  // (operator launches: tiles of exactly 3 / 5 / 6 / 7 rows too)
      case 3: return launch_inst<BITS, 3, kWaves, 0, LIN>(a, stream);
      case 5: return launch_inst<BITS, 5, kWaves, 0, LIN>(a, stream);
      case 6: return launch_inst<BITS, 6, kWaves, 0, LIN>(a, stream);
      case 7: return launch_inst<BITS, 7, kWaves, 0, LIN>(a, stream);
      default: break;
    }
  }
  if constexpr (BITS == 4 && !LIN) {
    if (batch == 1) {  // every K slice of the launch at most two steps per wave (o_proj): the kernel whose chunks are two steps (sqllm_fused.h: SHORT)
      bool brief = true;
      for (int i = 0; i < a.ga.n_seg; ++i) brief = brief && a.ga.seg[i].gm.units_per_wg <= 2 * kWaves * 4;
      if (brief) return launch_inst<BITS, 1, kWaves, 0, LIN, true>(a, stream);
    }
  }
  switch (batch_tile(batch)) {
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
  dim3 grid(a.ga.block0[a.ga.n_seg], (gm.batch + BT - 1) / BT);
  auto kern = sqllm_fused_cols<BITS, BT, kWaves>;
  const float* x = static_cast<const float*>(a.x);
  if (a.ev_start || a.ev_stop) hipExtLaunchKernelGGL(kern, grid, dim3(kWaves * 64), 0, stream, a.ev_start, a.ev_stop, 0, x, a.ga);
  else hipLaunchKernelGGL(kern, grid, dim3(kWaves * 64), 0, stream, x, a.ga);
  return hipGetLastError();
}

template <int BITS>
static hipError_t launch_cols_bits(const LaunchArgs& a, hipStream_t stream) {
  switch (batch_tile_op(a.ga.seg[0].gm.batch)) {  // (passes of exactly 3 / 5 / 6 rows too: round 6)
    case 1: return launch_cols_inst<BITS, 1>(a, stream);
    case 2: return launch_cols_inst<BITS, 2>(a, stream);
    case 3: return launch_cols_inst<BITS, 3>(a, stream);
    case 4: return launch_cols_inst<BITS, 4>(a, stream);
    case 5: return launch_cols_inst<BITS, 5>(a, stream);
    case 6: return launch_cols_inst<BITS, 6>(a, stream);
    case 7: return launch_cols_inst<BITS, 7>(a, stream);
    default: return launch_cols_inst<BITS, 8>(a, stream);
  }
}

// 1..kMaxSegments ops over one vec (a.ga), operator ABI, small batches: lane = column, vec from SGPRs
hipError_t launch_batched_cols(int bits, const LaunchArgs& a, hipStream_t stream) {
  return bits == 4 ? launch_cols_bits<4>(a, stream) : launch_cols_bits<3>(a, stream);
}

// the CSR and top-X terms of one wide-batch op (a.ga.seg[0]); a.xT = transposed vec or null
hipError_t launch_batched_sparse(const LaunchArgs& a, hipStream_t stream) {
  const KernelGeom& gm = a.ga.seg[0].gm;
  if (gm.csr_blocks + gm.topx_blocks <= 0) return hipSuccess;
  // blocks of 128 rows (two per lane in the CSR walk: fewer, longer workgroups) only where that leaves the chip several
  // rounds of them -- at 128 rows it halved a grid of 662 and the launch went from 61 to 100 us
  const int blocks = gm.csr_blocks + gm.topx_blocks;
  const int pass_rows = (long long)blocks * ((gm.batch + kSparsePassRows - 1) / kSparsePassRows) >= 2048 ? kSparsePassRows : 64;
  dim3 grid(blocks, (gm.batch + pass_rows - 1) / pass_rows);
  auto kern = sqllm_sparse_batched<kWaves>;
  const float* x = static_cast<const float*>(a.x);
  if (a.ev_start || a.ev_stop) hipExtLaunchKernelGGL(kern, grid, dim3(kWaves * 64), 0, stream, a.ev_start, a.ev_stop, 0, x, a.ga, a.xT, a.Bp, pass_rows);
  else hipLaunchKernelGGL(kern, grid, dim3(kWaves * 64), 0, stream, x, a.ga, a.xT, a.Bp, pass_rows);
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

// Small batches (2..16 rows): xT[k][rp], rp = the batch rounded up to a power of two, rows past the batch zero.  Read by
// the folded CSR walk and the top-X slabs of the fused small launch (csr_tile_fold_staged, topx_role_xt).  The kernel
// sits in front of every such launch, so its own round trips count: a thread = (batch row, four consecutive k's) -- ONE
// 16-byte load (a wave reads a whole line of each of its rows) and four 4-byte stores that a wave lays down as whole
// lines.  (One thread per k with a load per batch row: 5.6 us per launch in the kernel trace, profiles/r05_kt_13b_rows.summary.txt.)
__global__ void __launch_bounds__(256) sqllm_transpose_small(const float* __restrict__ x, float* __restrict__ xT, int batch, int K, int lr) {
  const unsigned t = blockIdx.x * 256u + threadIdx.x;
  const unsigned rp = 1u << lr;
  const unsigned b = t & (rp - 1u);
  const unsigned k = (t >> lr) * 4u;
  if (k >= (unsigned)K) return;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (b < (unsigned)batch) v = *reinterpret_cast<const f32x4*>(x + (size_t)b * K + k);  // (K is a multiple of 32)
  float* dst = xT + ((size_t)k << lr) + b;
  dst[0] = v.x;
  dst[rp] = v.y;
  dst[2 * rp] = v.z;
  dst[3 * rp] = v.w;
}

// The same in front of the fused small launch when it also takes vec ALREADY SPLIT: besides xT, vec as three bf16 planes
// in fragment order (the layout of sqllm_split_vec, row block 0 only: chunk ((kb * 3 + plane) * 64 + lane) * 16 bytes,
// lane = 16 * ((k / 8) % 4) + row, kb = k / 32; k block K / 32 all zero -- the address of lanes past a K range).  The
// dense role then loads its A fragments ready-made: the split in registers was 54 of the 152 vector instructions of a
// phase (2048 weights) of its loop.  A thread = (batch row, eight consecutive k's): two 16-byte loads, three 16-byte
// stores that a wave lays down as 1 KB runs, eight 4-byte stores into xT (whole lines per wave).
__global__ void __launch_bounds__(256) sqllm_prepare_small(const float* __restrict__ x, float* __restrict__ xT, u32x4* __restrict__ planes,
                                                           int batch, int K, int lr) {
  const unsigned t = blockIdx.x * 256u + threadIdx.x;
  const unsigned b = t & 15u;
  const unsigned k8 = t >> 4;  // group of eight k's
  const unsigned k = 8u * k8;
  if (k >= (unsigned)K + 32u) return;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (k < (unsigned)K && b < (unsigned)batch) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + (size_t)b * K + k), c = *reinterpret_cast<const f32x4*>(x + (size_t)b * K + k + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
  }
  uint32_t h[4], m[4], l[4];
  split8(v, h, m, l);
  u32x4* dst = planes + ((size_t)(k8 >> 2) * 3u * 64u + 16u * (k8 & 3u) + b);
  dst[0] = u32x4{h[0], h[1], h[2], h[3]};
  dst[64] = u32x4{m[0], m[1], m[2], m[3]};
  dst[128] = u32x4{l[0], l[1], l[2], l[3]};
  const unsigned rp = 1u << lr;
  if (xT && k < (unsigned)K && b < rp) {
    float* d = xT + ((size_t)k << lr) + b;
#pragma unroll
    for (int j = 0; j < 8; ++j) d[(size_t)j << lr] = v[j];
  }
}

hipError_t prepare_small(const float* x, float* xT, void* planes, int batch, int K, hipStream_t stream, hipEvent_t ev_start) {
  const int lr = batch <= 2 ? 1 : batch <= 4 ? 2 : batch <= 8 ? 3 : 4;
  const unsigned threads = (unsigned)(K / 8 + 4) * 16u;
  const dim3 grid((threads + 255u) / 256u);
  u32x4* pl = static_cast<u32x4*>(planes);
  if (ev_start) hipExtLaunchKernelGGL(sqllm_prepare_small, grid, dim3(256), 0, stream, ev_start, nullptr, 0, x, xT, pl, batch, K, lr);
  else hipLaunchKernelGGL(sqllm_prepare_small, grid, dim3(256), 0, stream, x, xT, pl, batch, K, lr);
  return hipGetLastError();
}

hipError_t transpose_small(const float* x, float* xT, int batch, int K, hipStream_t stream, hipEvent_t ev_start) {
  const int lr = batch <= 2 ? 1 : batch <= 4 ? 2 : batch <= 8 ? 3 : 4;
  const unsigned threads = (unsigned)(K / 4) << lr;
  const dim3 grid((threads + 255u) / 256u);
  if (ev_start) hipExtLaunchKernelGGL(sqllm_transpose_small, grid, dim3(256), 0, stream, ev_start, nullptr, 0, x, xT, batch, K, lr);
  else hipLaunchKernelGGL(sqllm_transpose_small, grid, dim3(256), 0, stream, x, xT, batch, K, lr);
  return hipGetLastError();
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
  if (g_fused_variant) {
    hipError_t err = hipSuccess;
    if (g_fused_variant(bits, a, stream, &err)) return err;
  }
  if (a.linear) return bits == 4 ? launch_bt<4, true>(a, stream) : launch_bt<3, true>(a, stream);
  return bits == 4 ? launch_bt<4, false>(a, stream) : launch_bt<3, false>(a, stream);
}

}  // namespace sqllm
