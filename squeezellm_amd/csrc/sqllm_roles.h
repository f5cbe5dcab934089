// sqllm_roles.h -- the sparse roles of a launch (CSR chunks balanced by nnz; top-X row slabs), shared
// by the fused kernels of sqllm_kernels.hip and the streaming kernel of sqllm_stream.hip.
// Reference arithmetic: squeezellm/quant_cuda_kernel.cu:1040-1089 (SPMV_ATOMIC[_BATCHED]),
// :1092-1164 (DenseMatVecKernel[Batched]).
#pragma once
#include "sqllm_decode.h"
#include "sqllm_probe.h"

namespace sqllm {

// Hook of the dependency-gated pass (sqllm_pass.hip): called by every thread of the workgroup, once, right before the
// role's first read of vec; the default does nothing.
struct NoGate {
  __device__ __forceinline__ void operator()() const {}
};

// One wave's share of a chunk in transposed-vec mode at 32 rows or fewer (see csr_role): st = the wave's WN
// non-zeros in LDS as [column | value bits | local row] planes kCsrChunk apart; R rows per pass -> 64 / R lane
// groups, each walking a contiguous run of WN * R / 64 non-zeros; row sums go to tile[row][local column] with an
// LDS add each (1.25 ns per active lane -- cheaper than the code a plain-store variant would add to every step).
template <int R, int WN>
__device__ __forceinline__ void xt_walk(const int* st, const float* __restrict__ xT, int Bp, int b0, int nb, float* tile, int TS, int lane) {
  constexpr int G = 64 / R, L = WN / G;
  constexpr int U = L < 32 ? L : 32;  // loads in flight per lane: the role is latency-bound
  const int bl = lane % R;
  const int base = (lane / R) * L;
  const float* xl = xT + (b0 + (bl < nb ? bl : 0));
  float* trow = tile + bl * TS;
  float acc = 0.f;
#pragma unroll 1
  for (int s0 = 0; s0 < L; s0 += U) {
    if (__builtin_amdgcn_readfirstlane(st[2 * kCsrChunk + s0]) < 0) break;  // (valid non-zeros are a prefix: group 0 has run out, so have all)
    float xv[U];
    int rr[U + 1];  // local rows, read before the loop below: its LDS adds would otherwise force every one to be re-read
#pragma unroll
    for (int u = 0; u < U; ++u) xv[u] = xl[(size_t)st[base + s0 + u] * Bp];  // (columns past the end are clamped re-reads, their values 0)
#pragma unroll
    for (int u = 0; u < U; ++u) rr[u] = st[2 * kCsrChunk + base + s0 + u];
    rr[U] = s0 + U == L ? -2 : st[2 * kCsrChunk + base + s0 + U];  // the run ends: whatever is open leaves
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc = __builtin_fmaf(__builtin_bit_cast(float, st[kCsrChunk + base + s0 + u]), xv[u], acc);
      if (rr[u] != rr[u + 1]) {
        if (rr[u] >= 0) atomicAdd(trow + rr[u], acc);
        acc = 0.f;
      }
    }
  }
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
//   then, LDS and registers only: the row searches of a lane's non-zeros run in lockstep in the staged
//   pointers; a lane holds a RUN of consecutive non-zeros, folds its own products and the wave sums the
//   lanes' open row segments with one DPP segmented scan per batch row; each row segment leaves with an
//   LDS add from the lane that holds its last non-zero, and each touched row as one global atomic.
//   (Wide batches read a transposed copy of vec instead -- lane = batch row, xt_walk above or the scalar
//   walk inside.)
// ------------------------------------------------------------------------------------------------
// CH: non-zeros per workgroup -- kCsrChunk, or (batch-1 operator launches that exceed the resident slots: sqllm_capi.hip, widen_csr_chunks) twice
// that: half as many, twice as deep workgroups in front of a multi-round grid.
template <int T, int BT, typename XT, typename AT, bool XTMODE = false, bool XCOH = false, typename GATE = NoGate, int CH = kCsrChunk>
__device__ __forceinline__ void csr_role(const XT* x, AT* __restrict__ y,
                                         const int* __restrict__ rows, const int* __restrict__ cols,
                                         const float* __restrict__ vals, int nnz, int K, int N, int b0,
                                         int nb, int chunk, float* lds, const Segment* lin, int lin_or_abl_bits = 0,
                                         const float* __restrict__ xT = nullptr, int Bp = 0,
                                         unsigned long long* tl = nullptr, GATE gate = GATE()) {
  constexpr bool LIN = sizeof(AT) == 8;
  int tid_ = threadIdx.x;
  // (inside the persistent pass kernel the role runs in a loop over work items: what it derives from the thread id
  // is recomputed per item -- hoisted out of that loop it would be live across every other role of the kernel)
  if constexpr (XCOH) asm volatile("" : "+v"(tid_));
  const int tid = tid_;
  static_assert(!XTMODE || CH == kCsrChunk, "the transposed-vec walks stage their planes kCsrChunk apart");
  static_assert(CH == kCsrChunk || sizeof(AT) == 4, "the fused linear counts a row's chunks in units of kCsrChunk");
  const int e0 = chunk * CH;
  int e1 = e0 + CH;
  if (e1 > nnz) e1 = nnz;
  if (e0 >= e1) return;
  // measurement library only (sqllm_probe.h: the bits are the constant 0 in the product and all of this folds away):
  // 1 = skip the role, 2 = skip the flush, 4 = skip the accumulation, 8 = no x gathers, bits 4.. = hold the role back
  const int cabl = SQLLM_ABLATION_BITS(lin_or_abl_bits);
  if (cabl & 1) return;
  for (int d = cabl >> 4; d > 0; --d) __builtin_amdgcn_s_sleep(8);  // (by (cabl >> 4) x ~0.2 us)
  // timeline probe (tools/timeline.py): the entry stamp is stored NEGATED, which tells a chunk workgroup from a dense one
  SQLLM_PROBE_NEG(tl, 0, tid == 0);
#define SQLLM_CSR_STAMP(I) SQLLM_PROBE(tl, I, tid == 0);

  // ---- round 1 ----
  constexpr int EPT = CH / T;  // non-zeros per thread
  static_assert(EPT >= 2, "a lane holds a run of consecutive non-zeros");
  int col[EPT];
  float val[EPT];
  // element of (thread, i): transposed-vec mode -- EPT runs of 64 that are consecutive within a wave (so that
  // only a wave's first and last row are shared with its neighbours)
  //  -- otherwise -- EPT consecutive non-zeros per lane (a wave holds 64 * EPT consecutive ones: the lane folds
  // its own run serially, so ONE segmented scan per batch row serves the whole run)
  auto elem = [&](int i) { return XTMODE ? e0 + (tid >> 6) * (64 * EPT) + 64 * i + (tid & 63) : e0 + EPT * tid + i; };
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    int e = elem(i);
    if (e > e1 - 1) e = e1 - 1;  // clamped re-read; masked below
    col[i] = cols[e];
    val[i] = vals[e];
  }
  // The span of row pointers the chunk needs, rows[c_lo .. c_hi] with rows[c_lo] <= e0 and rows[c_hi] > e1 - 1: one
  // sampled probe of `rows` per thread and two block-wide counts.  (A per-chunk table of these spans, written once
  // per matrix and handed in through the op descriptor, was built and measured: the chunk workgroups live 0.5-0.9 us
  // shorter -- their row pointers go out in the first round -- and the 7B s45 launches get 1-2.5 % SLOWER, the
  // earlier gathers landing in the dense workgroups' load phase; profiles/r03_timeline_csr.txt.  Not kept.)
  const int S = (N + T) / T;  // sample stride: T samples cover rows[0 .. N]
  const int si = tid * S;
  const int probe = rows[si < N ? si : N];
  // rows is non-decreasing with rows[0] = 0, so both predicates are true for a prefix of samples
  const int cnt_lo = __syncthreads_count(si <= N && probe <= e0);
  const int cnt_hi = __syncthreads_count(si <= N && probe <= e1 - 1);
  const int c_lo = (cnt_lo > 0 ? cnt_lo - 1 : 0) * S;  // rows[c_lo] <= e0
  int c_hi = cnt_hi * S;                                // rows[c_hi] > e1 - 1 (or the end)
  if (c_hi > N) c_hi = N;
  SQLLM_CSR_STAMP(1)  // round 1 (cols / vals / probes) has landed, both counts done
  const int n = c_hi - c_lo + 1;  // staged row pointers rows[c_lo .. c_hi]; candidate rows: n - 1
  const bool in_lds = n <= kCsrSpanMax;

  // ---- round 2 ----
  int* srows = reinterpret_cast<int*>(lds);  // [kCsrSpanMax]
  float* sacc = lds + kCsrSpanMax;           // [kCsrSpanMax]
  // the gather goes out first: the staging loop below waits for its own loads before it stores
  float xg[EPT];
  gate();  // (gated pass: vec is not read before the producing group is complete; cols / vals / probes are already here)
#pragma unroll
  for (int i = 0; i < EPT; ++i) xg[i] = XTMODE ? 0.f : ld_x<XCOH>(x + (size_t)b0 * K + col[i]);  // first batch row's gather
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
  // (passes of up to 128 rows, two per lane, where a 128-row tile fits -- a chunk spanning at most kCsrXtSpan / 2 rows;
  // otherwise 64 rows at a time)
  const bool two_rows = use_tile && nb > 64 && n <= kCsrXtSpan / 2;
  if (use_tile) for (int i = tid; i < (nb <= 16 ? 16 : nb <= 32 ? 32 : two_rows ? 128 : 64) * TS; i += T) tile[i] = 0.f;  // (the rows of this pass's lane groups)
  __syncthreads();
  SQLLM_CSR_STAMP(2)  // row pointers staged

  // local row of each non-zero: largest i with rows[c_lo + i] <= e.  The searches of a lane's non-zeros advance
  // in lockstep through ONE loop of a fixed, workgroup-uniform trip count (their LDS reads are independent and
  // overlap; once a search has converged, mid == lo and the step is a no-op): a while loop per non-zero is a
  // chain of 6-7 dependent LDS round trips EACH, 0.6-0.7 us of the workgroup's life per non-zero of a lane
  // (measured through the chunk size: 4 per lane added 1.3 us to the o_proj launch, profiles/r03_csr_chunk_topx_slab.txt)
  int lr[EPT];
  {
    int lo[EPT], hi[EPT];
    const int H = n - 1 > 1 ? n - 1 : 1;  // answer in [0, H): rows[c_lo + n - 1] > e by construction
#pragma unroll
    for (int i = 0; i < EPT; ++i) { lo[i] = 0; hi[i] = H; }
    if (in_lds) {
      const int iters = H > 1 ? 32 - __builtin_clz(H - 1) : 0;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
          const int mid = (lo[i] + hi[i]) >> 1;
          if (srows[mid] <= elem(i)) lo[i] = mid; else hi[i] = mid;
        }
      }
    } else {  // a chunk spanning > kCsrSpanMax rows (extremely sparse region): search in global memory
#pragma unroll
      for (int i = 0; i < EPT; ++i) {
        while (hi[i] - lo[i] > 1) {
          const int mid = (lo[i] + hi[i]) >> 1;
          if (rows[c_lo + mid] <= elem(i)) lo[i] = mid; else hi[i] = mid;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < EPT; ++i) lr[i] = (elem(i) < e1) ? lo[i] : -1;
  }
  SQLLM_CSR_STAMP(3)  // rows found
  if constexpr (XTMODE) {
    // Wide batches with a TRANSPOSED copy of vec (xT[k][row], written by sqllm_transpose_vec just before this
    // launch): lane = batch row, so a non-zero costs ONE coalesced read of xT[k] for all the rows instead of one
    // gather per row, 4 K bytes apart.  A pass of at most 16 / 32 rows splits the wave into G = 4 / 2 lane groups,
    // and each group walks its own run of the wave's non-zeros one at a time -- G non-zeros per step.  (With every lane a batch row whatever the batch, 9-16
    // rows paid 128 serial steps per wave in four latency-bound batches of loads with 48 of 64 lanes idle: the
    // sparse launch of a 16-row 13B op took 17 us; profiles/r03_wide_sparse_groups.txt.)  Columns, values and
    // local rows reach the groups through LDS (staged by the lanes that loaded them; same-wave traffic, in
    // order, no barrier).  At the last non-zero of a row the lanes park their sums in tile[row][column]: a
    // plain store, or an LDS add for a group's first row and for whatever it holds at the end of its run
    // (which the neighbouring groups / waves may hold parts of).  The tile leaves with the lanes along the
    // COLUMNS: coalesced atomics (lanes along the rows would hit one cache line each: measured 2.1 ms of a
    // 4.6 ms launch at 2048 rows).
    constexpr int WN = 64 * EPT;  // non-zeros per wave
    const int lane = tid & 63;
    int* st = reinterpret_cast<int*>(lds + kCsrSpanMax + 64 * (kCsrXtSpan + 1)) + (tid >> 6) * WN;  // [3][kCsrChunk]
    float* yf = reinterpret_cast<float*>(y);
    // what the one-group walks need, the same for every pass: where rows end, which lanes hold a non-zero
    unsigned long long ends[EPT], valid[EPT];
    int n_valid = 0;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int above = dpp_i32<0x130, 0xf>(lr[i], -2);  // the lane above's row (lane 63: none)
      ends[i] = __ballot(above != lr[i] && lr[i] >= 0);
      valid[i] = __ballot(lr[i] >= 0);
      n_valid += __builtin_popcountll(valid[i]);
    }
#pragma unroll
    for (int i = 0; i + 1 < EPT; ++i)  // a row that runs on into the next run keeps its sum in the register
      if ((valid[i + 1] & 1ull) && __builtin_amdgcn_readlane(lr[i], 63) == __builtin_amdgcn_readlane(lr[i + 1], 0)) ends[i] &= ~(1ull << 63);
    // The rows this workgroup was given (sqllm_sparse_batched: up to 128) go through in passes of `step`; the chunk's
    // columns, values and row searches above are paid once.
    const int step = two_rows ? 128 : 64;
    bool staged = false;
    for (int pb = 0; pb < nb; pb += step) {
      const int nbp = nb - pb < step ? nb - pb : step;
      const int bp0 = b0 + pb;
      if (pb) {  // the tile again (the previous pass's flush has read it)
        __syncthreads();
        if (use_tile) for (int i = tid; i < step * TS; i += T) tile[i] = 0.f;  // (the whole pass's rows: 128 of them in a two-rows-per-lane pass)
        __syncthreads();
      }
      if (nbp <= 32 && use_tile && !staged) {
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
          st[64 * i + lane] = col[i];
          st[kCsrChunk + 64 * i + lane] = lr[i] >= 0 ? __builtin_bit_cast(int, val[i]) : 0;
          st[2 * kCsrChunk + 64 * i + lane] = lr[i];
        }
        staged = true;
      }
      if (nbp <= 16 && use_tile) xt_walk<16, WN>(st, xT, Bp, bp0, nbp, tile, TS, lane);
      else if (nbp <= 32 && use_tile) xt_walk<32, WN>(st, xT, Bp, bp0, nbp, tile, TS, lane);
      else if (nbp > 64) {
        // 65-128 rows, TWO per lane (rows 2 lane, 2 lane + 1: one 8-byte read per non-zero): the walk is bound by its
        // per-non-zero instructions -- a scalar chain from v_readlane to the load's address -- not by bytes (reading
        // half-width values changed nothing: 412 vs 426 us per 13B gate/up op at 2048 rows), so each step now serves
        // twice the rows.
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        const int r0 = 2 * lane;
        const float* xl = xT + (bp0 + (r0 < nbp ? r0 : 0));  // (rows past the pass: the first pair again; never stored)
        float acc0 = 0.f, acc1 = 0.f;
        bool first_seg = true;
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
          constexpr int U = 32;
          for (int j0 = 0; j0 < 64; j0 += U) {
            if (((valid[i] >> j0) & 1ull) == 0) break;  // (valid lanes are a prefix)
            f32x2_t xv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
              const int k = __builtin_amdgcn_readlane(col[i], j0 + u);
              xv[u] = *reinterpret_cast<const f32x2_t*>(xl + (size_t)k * Bp);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
              const int j = j0 + u;
              const float v = ((valid[i] >> j) & 1ull) ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, val[i]), j)) : 0.f;
              acc0 = __builtin_fmaf(v, xv[u].x, acc0);
              acc1 = __builtin_fmaf(v, xv[u].y, acc1);
              if ((ends[i] >> j) & 1ull) {
                const int r = __builtin_amdgcn_readlane(lr[i], j);
                float* slot = tile + r0 * TS + r;
                if (first_seg || 64 * i + j == n_valid - 1) {  // the wave's first and last row: shared with its neighbours
                  atomicAdd(slot, acc0);
                  atomicAdd(slot + TS, acc1);
                } else {
                  slot[0] = acc0;
                  slot[TS] = acc1;
                }
                first_seg = false;
                acc0 = acc1 = 0.f;
              }
            }
          }
        }
      } else {
        // 33-64 rows (or a chunk spanning too many rows for the tile): one group, every lane a batch row.  Column, value and row come out of the owning lane with
        // v_readlane, so control flow and addresses are scalar (2 % faster at 2048 rows than the walk through LDS).
        const bool row_ok = lane < nbp;
        const float* xl = xT + (bp0 + (row_ok ? lane : 0));
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
                  if (first_seg || 64 * i + j == n_valid - 1) atomicAdd(slot, acc);  // the wave's first and last row: shared with its neighbours
                  else *slot = acc;
                } else if (row_ok) {
                  acc_add(yf + (size_t)(bp0 + lane) * N + c_lo + r, acc);
                }
                first_seg = false;
                acc = 0.f;
              }
            }
          }
        }
      }
      SQLLM_CSR_STAMP(4)  // this wave's non-zeros walked
      if (use_tile) {
        __syncthreads();
        const int nm1 = n - 1;
        for (int idx = tid; idx < nm1 * nbp; idx += T) {
          const int b = idx / nm1, r = idx - b * nm1;
          const float sum = tile[b * TS + r];
          if (sum != 0.f) acc_add(reinterpret_cast<float*>(y) + (size_t)(bp0 + b) * N + c_lo + r, sum);
        }
      }
    }
    SQLLM_CSR_STAMP(5)  // atomics issued
  } else {
  const int nm1 = n - 1 > 0 ? n - 1 : 1;
  // A wave holds 64 * EPT consecutive non-zeros, EPT per lane, i.e. a few whole or partial rows.  Per batch row:
  // the lane folds its own products serially (q[i] = running sum of the row segment that non-zero i belongs to),
  // ONE segmented inclusive scan across the lanes sums every lane's open (last) segment (row shifts 1 / 2 / 4 / 8,
  // then the row broadcasts 15 and 31 -- all DPP, no LDS traffic), and a row segment leaves from the lane that
  // holds its last non-zero.  The structure is the same for every batch row and is worked out once:
  //   b[i]      : a row ends between the lane's non-zeros i and i + 1
  //   take[d]   : scan step d adds the partial sum it is offered (the source lane's last row == this lane's;
  //               rows are sorted, so everything in between is that row too)
  //   flush_head: the lane holds a row end; its FIRST segment then closes what may have come in from the lane
  //               below (that lane's inclusive sum, if cont_prev)
  //   mid[i]    : a segment that starts and ends inside the lane (rows of 1-2 non-zeros), closing at i
  //   flush_tail: the lane above starts another row (or there is none)
  // (Before: non-zeros interleaved across the workgroup and one six-step ds_bpermute scan per non-zero and batch
  // row -- profiles/r03_sparse_role_batch.txt, r03_ab_csr_lane_runs.txt.  64 LDS adds colliding on 2-3
  // addresses, the first version, execute one lane at a time: 17 of 49 us; with the 2-3 lanes that are active
  // here an LDS float add costs 3-4 ns, profiles/r03_lds_atomic_sparse.txt.)
  const int rf = lr[0], rl = lr[EPT - 1];
  const bool bnd = rf != rl;
  bool b[EPT - 1], mid[EPT - 1];
  {
    bool any = false;
#pragma unroll
    for (int i = 0; i < EPT - 1; ++i) {
      b[i] = lr[i] != lr[i + 1];
      mid[i] = b[i] & any & (lr[i] >= 0);
      any |= b[i];
    }
  }
  bool take[6];
  take[0] = dpp_i32<0x111, 0xf>(rl, -2) == rl;
  take[1] = dpp_i32<0x112, 0xf>(rl, -2) == rl;
  take[2] = dpp_i32<0x114, 0xf>(rl, -2) == rl;
  take[3] = dpp_i32<0x118, 0xf>(rl, -2) == rl;
  take[4] = dpp_i32<0x142, 0xa>(rl, -2) == rl;
  take[5] = dpp_i32<0x143, 0xc>(rl, -2) == rl;
  // (every cross-lane read is a statement of its own, ahead of the logic: behind a short-circuit && it would run
  // with the lanes that fail the first test switched off -- and a DPP read of a switched-off lane returns `old`)
  const int prev_rl = dpp_i32<0x138, 0xf>(rl, -2), next_rf = dpp_i32<0x130, 0xf>(rf, -2);
  const bool cont_prev = prev_rl == rf;
  const bool flush_head = bnd & (rf >= 0);
  const bool flush_tail = (rl >= 0) & (next_rf != rl);
  for (int bs = 0; bs < nb; bs += g) {
    const int gb = nb - bs < g ? nb - bs : g;
    if (in_lds && bs > 0) {
      for (int i = tid; i < n * gb; i += T) sacc[i] = 0.f;
      __syncthreads();
    }
    // x of the group's rows for both non-zeros (unconditional loads: rows past the group re-read its last row)
    float xv[EPT][BT];
#pragma unroll
    for (int i = 0; i < EPT; ++i)
#pragma unroll
      for (int bb = 0; bb < BT; ++bb) {
        const int bi = bs + (bb < gb ? bb : gb - 1);
        xv[i][bb] = (bs == 0 && bb == 0) ? xg[i] : ld_x<XCOH>(x + (size_t)(b0 + bi) * K + col[i]);
        if (cabl & 8) xv[i][bb] = 1.f + bb;  // (measurement: no gathers)
      }
    const bool skip_acc = cabl & 4;
    if (!skip_acc) {
#pragma unroll
      for (int bb = 0; bb < BT; ++bb) {
        if (bb < gb) {
          float q[EPT];
          q[0] = val[0] * xv[0][bb];
#pragma unroll
          for (int i = 1; i < EPT; ++i) q[i] = __builtin_fmaf(val[i], xv[i][bb], b[i - 1] ? 0.f : q[i - 1]);
          float head = q[EPT - 1];
#pragma unroll
          for (int i = EPT - 2; i >= 0; --i) head = b[i] ? q[i] : head;  // the first row end wins
          float t = q[EPT - 1];  // the lane's open (last) segment
#define SQLLM_SCAN_STEP(D, CTRL, RM) { const float up = dpp_f32<CTRL, RM>(t); t += take[D] ? up : 0.f; }
          SQLLM_SCAN_STEP(0, 0x111, 0xf) SQLLM_SCAN_STEP(1, 0x112, 0xf) SQLLM_SCAN_STEP(2, 0x114, 0xf)
          SQLLM_SCAN_STEP(3, 0x118, 0xf) SQLLM_SCAN_STEP(4, 0x142, 0xa) SQLLM_SCAN_STEP(5, 0x143, 0xc)
#undef SQLLM_SCAN_STEP
          const float carry = dpp_f32<0x138, 0xf>(t);  // the lane below's inclusive sum
          if (flush_head) {
            const float h = head + (cont_prev ? carry : 0.f);
            if (in_lds) atomicAdd(sacc + bb * n + rf, h);
            else acc_add(y + (size_t)(b0 + bs + bb) * N + c_lo + rf, h);
          }
#pragma unroll
          for (int i = 1; i < EPT - 1; ++i) {
            if (mid[i]) {
              if (in_lds) atomicAdd(sacc + bb * n + lr[i], q[i]);
              else acc_add(y + (size_t)(b0 + bs + bb) * N + c_lo + lr[i], q[i]);
            }
          }
          if (flush_tail) {
            if (in_lds) atomicAdd(sacc + bb * n + rl, t);
            else acc_add(y + (size_t)(b0 + bs + bb) * N + c_lo + rl, t);
          }
        }
      }
    }
    if (in_lds) {
      __syncthreads();
      SQLLM_CSR_STAMP(4)  // x gathered, products scanned, row sums in LDS
      if (cabl & 2) continue;
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
            flag_nonfinite(y + at, sum);
            const unsigned target = (unsigned)lin->gm.k_slices + (unsigned)csr_chunks_of_row(r0, r1);
            column_done(*lin, y + at, atomicAdd(y + at, mine) + mine, target, at, c_lo + i);
          }
        } else {
          if (sum != 0.f) acc_add(y + at, sum);
        }
      }
      SQLLM_CSR_STAMP(5)  // atomics issued
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
// CSR rows FOLDED into the dense workgroups (the fused small launch of the split matrix-core kernel, 5..16 rows at 4
// bits, 9..16 at 3; replaces SPMV_ATOMIC_BATCHED, squeezellm/quant_cuda_kernel.cu:1061-1089, there).  CSR rows are
// output channels, so the non-zeros of a 64-column dense tile are ONE contiguous range of cols / vals,
// rows[col0] .. rows[col0 + 64); the workgroups that share the tile (its K slices / pieces) cut that range in
// proportion to their units of K, and each walks its share itself, after its dense loop: the sums meet the dense
// partial sums in LDS and leave in the tile's own epilogue.  No chunk workgroups (each of them 4-8 us of dependent
// memory round trips in a slot a dense workgroup could hold: with two 512-thread workgroups per CU they and the
// top-X slabs were the first round of the grid, the dense workgroups that found no slot a second one -- 50 us of a
// 13B s45 decoder layer at 8 rows, 80 at 16), no global atomics of their own, no search over rows[] (the tile's 65
// row pointers are staged with the codebooks).
//
// The walk: lane = (group, batch row); a group of R lanes (R = the batch rounded up to a power of two) takes a
// contiguous run of the share and walks it one non-zero per step -- column, local row and value come out of LDS (one
// address per group), the vec value is one load per lane: out of a TRANSPOSED copy of vec (xT[k][R], written by
// sqllm_transpose_small into the caller's workspace) the R lanes read one line per non-zero; without a workspace each
// lane gathers from its own row of vec, a line per non-zero AND batch row (~2.5 cycles of the CU's vector memory pipe
// each: 46 / 100 us of a 13B layer at 8 / 16 rows, profiles/r05_fold_staged_gathers.txt).  The running sum leaves
// with ONE LDS add per (CSR row, batch row) where the run crosses into the next row.
//   srp   LDS, rows[min(col0 + i, N)] for i = 0 .. 64 (staged by the caller, visible)
//   ssum  LDS [16][kFoldSumStride] floats, zero: sums by (batch row, local column)
//   stage LDS, 2 * kFoldStage words, free until the caller's epilogue (its cross-wave slabs): the share's columns
//         (+ local rows) and values, one coalesced round of loads for all of them -- the first PRE per thread are
//         loaded BEFORE the dense loop (pre_c / pre_v / pre_r: element tid + T i of the share)
// Measured on the way (profiles/r05_*): the walk as a chunk of exposed round trips at the end of every workgroup
// without staging (r05_fold_first_run), ONE wave walking beside seven decoding (r05_walker_wave: the walk is serial
// per lane group), 40 gathers per round trip (r05_walk_one_round), the batch tiles and the column-lane kernel folded
// the same way (2..4 rows: slower than their chunk role, which keeps that job).
// ------------------------------------------------------------------------------------------------
constexpr int kFoldRp = 128;   // staged row pointers (65 used; thread t stores entry t % 128)
constexpr int kFoldSumStride = kTileN + 1;  // floats between the batch rows of the sums: a group's lanes (one batch row each) add to different banks
constexpr int kFoldSum = 16 * kFoldSumStride;

// share of a tile's non-zeros [0, len) that belongs to units [0, u) of `units_total`: monotone, f(units_total) = len
__device__ __forceinline__ int fold_share(int len, int u, int units_total, unsigned inv) {
  return u >= units_total ? len : (int)__umulhi((unsigned)len, (unsigned)u * inv);
}

constexpr int kFoldStage = 4096;  // non-zeros staged per pass (a larger share: more passes)

// the share [sbeg, send) of this piece in the tile's non-zeros (absolute indices into cols / vals)
__device__ __forceinline__ void fold_piece_share(const int* srp, int u_beg, int u_end, int units_total, int* sbeg, int* send) {
  const int lo = srp[0];
  int len = srp[kTileN] - lo;
  if (len < 0) len = 0;
  const unsigned inv = 0xFFFFFFFFu / (unsigned)units_total;
  *sbeg = lo + fold_share(len, u_beg, units_total, inv);
  *send = lo + fold_share(len, u_end, units_total, inv);
}

// local rows of N non-zeros (absolute indices ea[]): the largest r with srp[r] <= ea, the N searches side by side
template <int N>
__device__ __forceinline__ void fold_rows_of(const int* srp, const int (&ea)[N], int (&r)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) r[i] = 0;
#pragma unroll
  for (int s = kTileN / 2; s > 0; s >>= 1) {
    int p[N];
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = srp[r[i] + s];
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] += p[i] <= ea[i] ? s : 0;
  }
}

// The walk itself (every thread of the workgroup must call it: barriers).  U = gathers of vec per lane and round trip;
// pre_r: the local rows of the preloaded non-zeros, six bits each (fold_rows_of, worked out before the dense loop as well)
template <int T, int U, int PRE>
__device__ __forceinline__ void csr_tile_fold_staged(const float* __restrict__ x, const float* __restrict__ xT, const int* __restrict__ cols,
                                                     const float* __restrict__ vals, int K, int b0, int nb, int sbeg, int send,
                                                     const int* srp, float* ssum, int* stage, int tid, const int (&pre_c)[PRE],
                                                     const float (&pre_v)[PRE], unsigned pre_r, unsigned long long* tl = nullptr) {
  // staged per non-zero: its column with its LOCAL CSR ROW in the top six bits (K < 2^26: the host routes wider
  // matrices elsewhere), and its value.  With the row at hand the walk has no search, no dependent read and no loop
  // at a row boundary: a step is compare / masked LDS add / select / FMA.  (The first version tracked the row end
  // with a data-dependent loop inside every step: ~45 instructions per step, 14 us of a 13B gate/up launch at 16 rows.)
  int* scol = stage;
  float* sval = reinterpret_cast<float*>(stage + kFoldStage);
  const int lr = nb <= 2 ? 1 : nb <= 4 ? 2 : nb <= 8 ? 3 : 4;  // log2 of the lanes per group
  const int bl = tid & ((1 << lr) - 1);
  const bool live = bl < nb;
  const float* xl = xT ? xT + bl : x + (size_t)(b0 + (live ? bl : nb - 1)) * K;
  const int csh = xT ? lr : 0;  // a column's offset: col << csh
  float* srow = ssum + bl * kFoldSumStride;
  for (int p0 = sbeg; p0 < send; p0 += kFoldStage) {  // (workgroup-uniform)
    const int np = send - p0 < kFoldStage ? send - p0 : kFoldStage;
    // ---- stage the pass's columns (+ rows) and values ----
    if (p0 == sbeg) {
#pragma unroll
      for (int i = 0; i < PRE; ++i) {
        if (tid + T * i < np) {
          scol[tid + T * i] = pre_c[i] | (((pre_r >> (6 * i)) & 63) << 26);
          sval[tid + T * i] = pre_v[i];
        }
      }
    }
    for (int i0 = (p0 == sbeg ? PRE * T : 0); i0 < np; i0 += 4 * T) {
      int cc[4];
      float vv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k * T + tid;
        const int ii = i < np ? i : np - 1;
        cc[k] = cols[p0 + ii];
        vv[k] = vals[p0 + ii];
      }
      int ea[4], rr[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) ea[k] = p0 + i0 + k * T + tid;
      fold_rows_of<4>(srp, ea, rr);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k * T + tid;
        if (i < np) {
          scol[i] = cc[k] | (rr[k] << 26);
          sval[i] = vv[k];
        }
      }
    }
    __syncthreads();
    SQLLM_PROBE(tl, 3, tid == 0);  // share staged
    // ---- walk: T >> lr groups, contiguous runs of the pass ----
    const int per = (np + (T >> lr) - 1) >> (__builtin_ctz(T) - lr);
    int e = (tid >> lr) * per;  // (index within the pass)
    int g_hi = e + per;
    if (g_hi > np) g_hi = np;
    if (e < g_hi) {
      int rprev = (unsigned)scol[e] >> 26;
      float acc = 0.f;
      auto step = [&](unsigned cr, float v, float xv) {
        const int r = cr >> 26;
        if (r != rprev) {
          if (live) atomicAdd(srow + rprev, acc);
          acc = 0.f;
        }
        rprev = r;
        acc = __builtin_fmaf(v, xv, acc);
      };
      // (Measured and dropped: 40 gathers per round trip, columns / rows / values re-read from the stage at their use --
      // one round trip for a typical run at 16 rows, but the range checks and second reads cost more than the round
      // trip saves: 149 -> 163 us per 13B layer at 16 rows, profiles/r05_walk_one_round.txt.)
      for (; e + U <= g_hi; e += U) {  // whole batches: no range checks
        unsigned cr[U];
        float xv[U], v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          cr[u] = (unsigned)scol[e + u];
          v[u] = sval[e + u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = xl[(size_t)(cr[u] & 0x3FFFFFFu) << csh];
#pragma unroll
        for (int u = 0; u < U; ++u) step(cr[u], v[u], xv[u]);
      }
      if (e < g_hi) {  // the run's last, partial batch (clamped re-reads of its last non-zero, skipped)
        unsigned cr[U];
        float xv[U], v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int ee = e + u < g_hi ? e + u : g_hi - 1;
          cr[u] = (unsigned)scol[ee];
          v[u] = sval[ee];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = xl[(size_t)(cr[u] & 0x3FFFFFFu) << csh];
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (e + u < g_hi) step(cr[u], v[u], xv[u]);
      }
      if (live) atomicAdd(srow + rprev, acc);
    }
    SQLLM_PROBE(tl, 4, tid == 0);  // wave 0's groups walked
    __syncthreads();  // the stage is free again (next pass, or the caller's slabs)
  }
}

// ------------------------------------------------------------------------------------------------
// top-X role: full_rows is fp32 [K, topX] row-major; a workgroup takes kTopxRows consecutive k's,
// i.e. one contiguous slab of kTopxRows*topX floats, and streams it coalesced (the reference keeps
// topX of 128 lanes busy with stride-topX reads, quant_cuda_kernel.cu:1113-1118).
// ------------------------------------------------------------------------------------------------
// RB: batch rows per pass (loads of all of them in flight together, one barrier pair per pass).  The batch-1 kernels
// use 1 (their register budget); the fused small launch 8 -- row by row, a 16-row launch kept its top-X workgroups (first
// in the grid) for 17 us, and the dense workgroups that found no slot beside them became a second round
// (profiles/r05_small_split_timeline.txt).
template <int T, typename XT, typename AT, bool XCOH = false, typename GATE = NoGate, int RB = 1>
__device__ __forceinline__ void topx_role(const XT* x, AT* __restrict__ y,
                                          const float* __restrict__ full_rows,
                                          const int* __restrict__ full_idx, int topX, int K, int N,
                                          int b0, int nb, int slab, float* lds, GATE gate = GATE()) {
  int tid_ = threadIdx.x;
  if constexpr (XCOH) asm volatile("" : "+v"(tid_));  // (see csr_role)
  const int tid = tid_;
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
    // two more barriers cost 0.6-0.9 us on the grouped 7B launches), one barrier pair per pass of RB batch rows.
    static_assert(T == 512 && kTopxRows % 32 == 0, "32 lane rows x NI k's cover the slab");
    static_assert(RB * (T / 64) * 16 <= kTopxLds && 16 * RB <= T, "a pass's partial sums: [RB][waves][16]");
    constexpr int NI = kTopxRows / 32;
    const int c = tid & 15, krow = tid >> 4;  // krow 0..31
    const int lane = tid & 63, wave = tid >> 6;
    const bool live = c < topX;
    float frv[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      int k = k0 + krow + 32 * i;
      if (k > k1 - 1) k = k1 - 1;  // clamped re-read, masked below
      frv[i] = live ? full_rows[(size_t)k * topX + c] : 0.f;
    }
    const int dst = live ? full_idx[c] : 0;
    gate();  // (gated pass: the slab of full_rows is in registers, vec comes after the gate)
    for (int bp = 0; bp < nb; bp += RB) {
      float p[RB];
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int b = bp + r < nb ? bp + r : nb - 1;  // (rows past the batch: the last one again; never added)
        const XT* xb = x + (size_t)(b0 + b) * K;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int k = k0 + krow + 32 * i;
          q = __builtin_fmaf(frv[i], k < k1 ? ld_x<XCOH>(xb + k) : 0.f, q);
        }
        p[r] = q;
      }
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        p[r] += __shfl_xor(p[r], 16, 64);
        p[r] += __shfl_xor(p[r], 32, 64);
      }
      if (bp > 0) __syncthreads();  // the previous pass's sums have been read
      if (lane < 16) {
#pragma unroll
        for (int r = 0; r < RB; ++r) lds[(r * (T / 64) + wave) * 16 + lane] = p[r];
      }
      __syncthreads();
      if (tid < 16 * RB) {
        const int r = tid >> 4, cc = tid & 15;
        if (cc < topX && bp + r < nb) {
          float sum = 0.f;
#pragma unroll
          for (int w = 0; w < T / 64; ++w) sum += lds[(r * (T / 64) + w) * 16 + cc];
          acc_add(y + (size_t)(b0 + bp + r) * N + dst, sum);  // (cc == c: this thread's own early-loaded index)
        }
      }
    }
    return;
  }
  const bool in_lds = topX <= kTopxLds;
  float* sacc = lds;
  gate();
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
      const float p = fr[e] * ld_x<XCOH>(xb + kk);
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
// top-X role of the fused small launch when vec comes TRANSPOSED (xT[k][2^lr], sqllm_transpose_small): the slabs
// [s0, s1) of kTopxRows k's each, all batch rows at once -- the 16 lanes of a lane row read ONE line per k for every
// batch row (row by row from vec: 16 x 8 scalar loads per thread and slab, 4.5 us per pass of 8 rows), the partial
// sums stay in registers across the slabs, one reduction at the end.  A workgroup therefore takes several slabs and
// an op needs only 8-16 such workgroups: few enough to sit beside ONE round of dense workgroups for the whole launch.
//   lds: 16 x (T / 64) x 16 floats
// ------------------------------------------------------------------------------------------------
template <int T>
__device__ __forceinline__ void topx_role_xt(const float* __restrict__ xT, int lr, float* __restrict__ y,
                                             const float* __restrict__ full_rows, const int* __restrict__ full_idx, int topX,
                                             int K, int N, int nb, int s0, int s1, float* lds) {
  static_assert(T == 512 && kTopxRows % 32 == 0, "32 lane rows x NI k's cover a slab");
  constexpr int NI = kTopxRows / 32;
  const int tid = threadIdx.x;
  const int c = tid & 15, krow = tid >> 4;
  const int lane = tid & 63, wave = tid >> 6;
  const bool live = c < topX;
  const int rp = 1 << lr;
  float p[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) p[r] = 0.f;
  for (int s = s0; s < s1; ++s) {
    const int k0 = s * kTopxRows;
#pragma unroll 4
    for (int i = 0; i < NI; ++i) {  // (four k's -- 20 loads -- in flight per thread: the role is a chain of round trips)
      const int k = k0 + krow + 32 * i;
      const int kc = k < K ? k : K - 1;  // clamped re-read, its weight zero
      const float f = (live && k < K) ? full_rows[(size_t)kc * topX + c] : 0.f;
      const float* xk = xT + ((size_t)kc << lr);
      if (rp == 2) {
        const f32x2 v = *reinterpret_cast<const f32x2*>(xk);
        p[0] = __builtin_fmaf(f, v.x, p[0]);
        p[1] = __builtin_fmaf(f, v.y, p[1]);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (4 * q < rp) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xk + 4 * q);
            p[4 * q] = __builtin_fmaf(f, v.x, p[4 * q]);
            p[4 * q + 1] = __builtin_fmaf(f, v.y, p[4 * q + 1]);
            p[4 * q + 2] = __builtin_fmaf(f, v.z, p[4 * q + 2]);
            p[4 * q + 3] = __builtin_fmaf(f, v.w, p[4 * q + 3]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    p[r] += __shfl_xor(p[r], 16, 64);
    p[r] += __shfl_xor(p[r], 32, 64);
  }
  if (lane < 16) {
#pragma unroll
    for (int r = 0; r < 16; ++r) lds[(r * (T / 64) + wave) * 16 + lane] = p[r];
  }
  __syncthreads();
  if (tid < 256) {
    const int r = tid >> 4, cc = tid & 15;
    if (cc < topX && r < nb) {
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < T / 64; ++w) sum += lds[(r * (T / 64) + w) * 16 + cc];
      acc_add(y + (size_t)r * N + full_idx[cc], sum);
    }
  }
}

#undef SQLLM_CSR_STAMP

}  // namespace sqllm
