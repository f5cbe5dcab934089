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
          flag_nonfinite(reinterpret_cast<u64*>(y) + at, sum);
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
  constexpr int NBUF = (BITS == 4) ? (BT <= 2 ? 4 : 2) : ((BT == 1 && !HALF) ? 2 : 1);
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
// the fused kernel
// ------------------------------------------------------------------------------------------------
// Occupancy is what this kernel lives on (measured, DESIGN.md 4.1: thread-level parallelism beats
// instruction-level parallelism here -- software-pipelining the stages at 104 VGPRs lost 8 %, while
// halving the live lookups won up to 18 %): the decode stages work on one column pair at a time
// (16 live lookups instead of 32), which lets the batch-1 kernels fit 64 VGPRs, i.e. FOUR 8-wave
// workgroups per CU; the wider batch tiles take what they need up to 128 (two per CU).
template <int BITS, int BT, int WAVES, int ABL, bool LIN>
__global__ void __launch_bounds__(WAVES * 64, (ABL & 64) ? 8 : ((BT == 1 && SQLLM_HALF_STAGES) ? 8 : ((BT == 2 || (BT == 4 && BITS == 4)) ? 6 : 4)))
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
                                         d, gm.col_tiles, gm.units_total, gm.units_per_wg, lds, sg, LIN ? &sg : nullptr);
  } else if (sp >= 0 && sp < gm.csr_blocks) {
    csr_role<T, BT, XT, AT>(x, reinterpret_cast<AT*>(sg.y), sg.rows, sg.cols, sg.vals, gm.nnz, gm.K, gm.N, b0, nb, sp, lds,
                        LIN ? &sg : nullptr, gm.sparse_last >> 1, nullptr, 0,
#ifdef SQLLM_ABLATION_BUILD
                        (!LIN && sg.bias) ? reinterpret_cast<unsigned long long*>(const_cast<float*>(sg.bias)) +
                                                8ull * (blockIdx.x + (unsigned long long)gridDim.x * blockIdx.y) : nullptr
#else
                        nullptr
#endif
    );
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
// 2048 rows.  Passes of 64 rows (blockIdx.y); blockIdx.x = CSR chunks, then top-X slabs.
//   xT != null: the CSR role reads the transposed copy of vec (lane = batch row);
//   xT == null (no scratch, or the stream is capturing): it gathers from vec, 32 rows at a time.
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
sqllm_sparse_batched(const float* x, const GroupArgs ga, const float* xT, int Bp) {
  constexpr int T = WAVES * 64;
  __shared__ __attribute__((aligned(16))) float lds[cmax(kCsrSpanMax + cmax(kCsrSpanMax, 64 * (kCsrXtSpan + 1) + 3 * kCsrChunk), kTopxLds)];
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
      csr_role<T, 1, float, float, true>(x, sg.y, sg.rows, sg.cols, sg.vals, gm.nnz, gm.K, gm.N, m0, rows_here, sp, lds, nullptr, 0, xT, Bp
#ifdef SQLLM_ABLATION_BUILD
                                         , sg.bias ? reinterpret_cast<unsigned long long*>(const_cast<float*>(sg.bias)) +
                                                         8ull * (blockIdx.x + (unsigned long long)gridDim.x * blockIdx.y) : nullptr
#endif
      );
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
    hipExtLaunchKernelGGL(kern, grid, dim3(WAVES * 64), a.lds_pad, stream, a.ev_start, a.ev_stop, 0, a.x, a.ga);
  } else {
    hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), a.lds_pad, stream, a.x, a.ga);
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
  dim3 grid(a.ga.block0[a.ga.n_seg], (gm.batch + BT - 1) / BT);
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

// 1..kMaxSegments ops over one vec (a.ga), operator ABI, small batches: lane = column, vec from SGPRs
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
