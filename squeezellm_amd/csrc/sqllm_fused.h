// sqllm_fused.h -- the fused batch-tile kernel as a TEMPLATE: dense tiles, CSR chunks and top-X slabs in one grid
// (sqllm_kernels.hip instantiates the product's variants; the measurement library instantiates its ablations of the
// same template in csrc/experimental/sqllm_ablation.hip).
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
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <type_traits>

#include "sqllm_kernels.h"

#include "sqllm_decode.h"
#include "sqllm_roles.h"

namespace sqllm {

// ------------------------------------------------------------------------------------------------
// Dense epilogue (shared by the dense-role variants): fold the 4 lane rows, then the waves through
// LDS, one atomic per column.  `slabs` = LDS area [WAVES][BT][64] floats followed by the ticket.
// ------------------------------------------------------------------------------------------------
template <int BT, int WAVES, int ABL>
__device__ __forceinline__ void dense_epilogue(const f32x2 (&acc)[2][BT], float* slabs, const float* topx_sum,
                                               bool fold_topx, float* __restrict__ y, int N, int col0, int b0,
                                               int nb, int lane, int wave, const Segment& sg, const Segment* lin,
                                               unsigned long long* tl /* timeline stamps: sqllm_probe.h (null in the product) */) {
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
          flag_nonfinite(reinterpret_cast<u64*>(y) + at, sum);
          total[b] = atomicAdd(reinterpret_cast<u64*>(y) + at, mine) + mine;
        } else {
          atomicAdd(y + at, sum);
        }
      }
    }
    SQLLM_PROBE(tl, 3, lane == 0);  // atomics issued by the combining wave
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
template <int BITS, int BT, int WAVES, int ABL, typename XT, bool HALF = false, bool SHORT = false>
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
  // steps per chunk (4-bit: even, steps pair up for x -- four at batch 1, two in the batch tiles: the 2-row tile fits 64 VGPRs with two;
  // 3-bit: 12 VGPRs of weights per step)
  // SHORT (4-bit batch 1, chosen per launch by the host: launch_bt): every K slice of the launch is at most TWO steps per wave (o_proj: 64 units) -- a chunk
  // of two then issues no loads it will not decode (the chunk of four re-read the slice's last unit twice): o_proj 4.83 -> 4.65 us (profiles/r06_short_chunks.txt)
  constexpr int NBUF = (BITS == 4) ? ((BT == 1 && !SHORT) ? 4 : 2) : ((BT == 1 && !HALF) ? 2 : 1);
  constexpr int NXR = (BITS == 4) ? NBUF / 2 : 2 * NBUF;  // x registers per chunk and batch row
  constexpr int STEP = WAVES * 4;                // units a workgroup step covers
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, grp = lane >> 4;
  // Dense waves above the sparse roles' waves in the SIMD's issue arbitration, where the host asked for it (sqllm_capi.hip:
  // set_dense_priority -- 3-bit batch-1 launches with sparse roles whose workgroups are all resident at once).
  if (sg.gm.dense_prio == 1) __builtin_amdgcn_s_setprio(1);
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
  // timeline probe (sqllm_probe.h; nothing in the product): wave 0 stamps entry / barrier passed / decode done, the
  // combining wave stamps the end
  unsigned long long* tl = lin ? nullptr : SQLLM_PROBE_PTR(sg);
  SQLLM_PROBE(tl, 0, tid == 0);
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
  SQLLM_PROBE(tl, 1, tid == 0);

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

  SQLLM_PROBE(tl, 2, tid == 0);
  SQLLM_PROBE(tl, 4 + (wave & 3), lane == 0);  // decode end of waves 0-3
  if constexpr (PAIR) {
    acc[0][0] = f32x2{accp[0].x + accp[0].y, accp[1].x + accp[1].y};
    acc[1][0] = f32x2{accp[2].x + accp[2].y, accp[3].x + accp[3].y};
  }
  dense_epilogue<BT, WAVES, ABL>(acc, lds + kCodebookFloats, topx_sum, fold_topx, y, N, col0, b0, nb, lane, wave, sg, lin, tl);
}

// ------------------------------------------------------------------------------------------------
// the fused kernel
// ------------------------------------------------------------------------------------------------
// Occupancy is what this kernel lives on (measured, DESIGN.md 4.1: thread-level parallelism beats
// instruction-level parallelism here -- software-pipelining the stages at 104 VGPRs lost 8 %, while
// halving the live lookups won up to 18 %): the decode stages work on one column pair at a time
// (16 live lookups instead of 32), which lets the batch-1 kernels fit 64 VGPRs, i.e. FOUR 8-wave
// workgroups per CU; the wider batch tiles take what they need up to 128 (two per CU).
// waves per SIMD the register allocation must leave room for (= 8-wave workgroups per CU x 2): batch 1 and the 4-bit 2-row tile four
// workgroups (64 VGPRs: half stages), the 3-bit 2-row, the 3-row and the 4-bit 4- / 5- / 6-row tiles three (80), everything wider two (128)
constexpr bool fused_half_stages(int bits, int bt) { return SQLLM_HALF_STAGES && (bt == 1 || (bt == 2 && bits == 4)); }
constexpr int fused_min_waves(int bits, int bt, int abl) {
  return (abl & 64) ? 8 : (fused_half_stages(bits, bt) ? 8 : ((bt == 2 || bt == 3 || (bt >= 4 && bt <= 6 && bits == 4)) ? 6 : 4));
}

template <int BITS, int BT, int WAVES, int ABL, bool LIN, bool SHORT = false>
__global__ void __launch_bounds__(WAVES * 64, fused_min_waves(BITS, BT, ABL))
sqllm_fused_matvec(const void* xv, const GroupArgs ga) {
  // Half stages (one column pair at a time) for batch 1 and -- round 6 -- the 4-bit 2-row tile: 63 VGPRs with two steps per chunk, the fourth
  // workgroup per CU (13B s45 at 2 rows: o_proj 8.3 -> 7.6 us, gate/up dense-only 24.0 -> 23.2; profiles/r06_tile2_half.txt).  The 3- / 4-row
  // tiles spill 52 / 126 registers at 64 and the 3-bit 2-row tile 5: they keep whole stages at three workgroups per CU.
  constexpr bool HALF = fused_half_stages(BITS, BT);
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
    dense_role<BITS, BT, WAVES, ABL, XT, HALF, SHORT>(x, reinterpret_cast<const u32x4*>(sg.q), sg.y, sg.lut, gm.K, gm.N, b0, nb,
                                         d, gm.col_tiles, gm.units_total, gm.units_per_wg, lds, sg, LIN ? &sg : nullptr);
  } else if (sp >= 0 && sp < gm.csr_blocks) {
    if (gm.dense_prio == 2) __builtin_amdgcn_s_setprio(1);  // (the sparse workgroups first out of the way: set_role_priority)
    if constexpr (BT <= 5 && !LIN) {  // (the 6-row tile would spill with the wide role in it)
      if (gm.csr_wide) {  // (half as many, twice as deep: widen_csr_chunks)
        csr_role<T, BT, XT, AT, false, false, NoGate, 2 * kCsrChunk>(x, reinterpret_cast<AT*>(sg.y), sg.rows, sg.cols, sg.vals, gm.nnz, gm.K, gm.N, b0, nb, sp, lds,
                                                                     nullptr, gm.sparse_last >> 1, nullptr, 0, SQLLM_PROBE_PTR(sg));
        return;
      }
    }
    csr_role<T, BT, XT, AT>(x, reinterpret_cast<AT*>(sg.y), sg.rows, sg.cols, sg.vals, gm.nnz, gm.K, gm.N, b0, nb, sp, lds,
                        LIN ? &sg : nullptr, gm.sparse_last >> 1, nullptr, 0, LIN ? nullptr : SQLLM_PROBE_PTR(sg));
  } else if (sp >= gm.csr_blocks && sp < gm.csr_blocks + gm.topx_blocks) {
    // (never taken when the plan folds the top-X rows into the dense tiles)
    if (gm.dense_prio == 2) __builtin_amdgcn_s_setprio(1);
    topx_role<T, XT, AT, false, NoGate, BT>(x, reinterpret_cast<AT*>(sg.y), sg.full_rows, sg.full_idx, gm.topX, gm.K, gm.N, b0, nb, sp - gm.csr_blocks, lds);  // (all BT rows of the pass in one go)
  }
}

template <int BITS, int BT, int WAVES, int ABL = 0, bool LIN = false, bool SHORT = false>
inline hipError_t launch_inst(const LaunchArgs& a, hipStream_t stream) {
  const int batch = a.ga.seg[0].gm.batch;
  dim3 grid(a.ga.block0[a.ga.n_seg], (batch + BT - 1) / BT);
  auto kern = sqllm_fused_matvec<BITS, BT, WAVES, ABL, LIN, SHORT>;
  if (a.ev_start || a.ev_stop) {
    // same kernel, with the dispatch's own begin/end timestamps exposed through two events
    hipExtLaunchKernelGGL(kern, grid, dim3(WAVES * 64), a.lds_pad, stream, a.ev_start, a.ev_stop, 0, a.x, a.ga);
  } else {
    hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), a.lds_pad, stream, a.x, a.ga);
  }
  return hipGetLastError();
}

}  // namespace sqllm
