// sqllm_pass.hip -- the dependency-gated pass: consecutive groups of batch-1 ops (a decode pass: q/k/v | o_proj |
// gate/up | down_proj | q/k/v of the next layer ...), each group reading what the previous one wrote, as ONE
// persistent launch instead of one launch per group.
//
// Why.  At batch 1 a group streams 8-47 MB of packed weights; as a launch of its own it costs
// t = 3.05 us + bytes / 4.8 TB/s (dispatch ramp, argument reads, codebooks and first bytes,
// the cross-wave combine, the drain, a cold instruction cache), and 128 launch boundaries per token put a third
// of the pass into that fixed part.  None of it depends on the previous group's result: only `vec` does.  So here
// the workgroups are long-lived and walk a list of work items (dense tile x K slice, CSR chunk, top-X slab) in
// pass order; a workgroup that runs out of items of group g goes straight on to an item of group g+1 -- reads
// its descriptor, fetches the tile's codebooks and its first chunk of packed weights into registers -- and only
// THEN waits for group g to be complete, right where `vec` is first consumed.  The reference's launch structure
// that this replaces: one to three dependent launches per op on the legacy stream
// (squeezellm/quant_cuda_kernel.cu:157-179, :510-577).
//
// Ordering between groups (everything else is read ahead of it: weights, codebooks and the sparse structure are
// constants of a pass):
//   producer  every work item ends with ONE device-scope atomic on its group's arrival counter, issued after
//             the item's own accumulations into `mul` -- device-scope fp32 atomics, performed at the memory
//             side -- have been acknowledged (s_waitcnt vmcnt(0));
//   consumer  one wave polls the counter of the group it depends on (relaxed agent-scope loads, s_sleep in
//             between), the workgroup barrier releases the other waves, and `vec` is then read with
//             agent-scope (sc1) loads, which this CU's L1 never serves.
// No fence on either side (MI355X_MICROARCH.md "Valid forms": agent atomics both sides; an agent acquire per
// item would cost 1.7-6.5 us).  Every gate is bounded: a launch whose gate stays shut for `timeout_ticks`
// raises status[0] and lets everything through (the results are then garbage, the launch still ends).
//
// Work items are taken from ONE queue in pass order (a device-scope ticket per item, fetched one item ahead): only
// workgroups that are actually running hold items, so every item below an item that waits at a gate is in the hands
// of a running workgroup -- whatever the chip admits.  (Dealing the items round-robin over a grid "that the chip
// holds at once" deadlocked as soon as another kernel ran beside the pass: with a second queue active the hardware
// kept exactly three of the four workgroups per CU resident and never admitted the rest while the first ones spun --
// profiles/r04_pass_residency.txt.)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "sqllm_pass.h"

#include "sqllm_decode.h"
#include "sqllm_roles.h"

namespace sqllm {

namespace {

// the pass tables are constants of the launch: read them through the constant address space (scalar loads, scalar
// cache) -- through a plain global pointer the atomics of the loop body would make them vector loads
template <typename T>
__device__ __forceinline__ T ld_const(const T* p) {
  static_assert(sizeof(T) % 4 == 0 && std::is_trivially_copyable<T>::value, "dword-wise copy");
  constexpr int n = sizeof(T) / 4;
  const __attribute__((address_space(4))) uint32_t* c =
      reinterpret_cast<const __attribute__((address_space(4))) uint32_t*>(reinterpret_cast<uintptr_t>(p));
  uint32_t w[n];
#pragma unroll
  for (int i = 0; i < n; ++i) w[i] = c[i];
  T t;
  __builtin_memcpy(&t, w, sizeof(T));
  return t;
}

// Pointers that come out of the pass tables are plain integers to the compiler: without this it cannot tell global
// from LDS / scratch and every access through them becomes a FLAT instruction (which also drags lgkmcnt into every
// wait on a vector load).
template <typename T>
__device__ __forceinline__ T* as_global(T* p) {
  // integer -> global pointer -> generic pointer: the address-space inference pass then knows where it points
  return (T*)reinterpret_cast<__attribute__((address_space(1))) T*>(reinterpret_cast<uintptr_t>(p));
}

__device__ __forceinline__ unsigned ld_agent_u32(const unsigned* p) {
  return __hip_atomic_load(SQLLM_GLOBAL(const unsigned, p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Sum of the group's arrival shards >= total?  Wave-uniform result; lanes 0..kPassShards-1 read one shard each.
__device__ __forceinline__ bool gate_is_open(const unsigned* gate, int total, int lane) {
  static_assert(kPassShards == 8, "three row shifts fold eight lanes");
  int v = lane < kPassShards ? (int)ld_agent_u32(gate + lane * kPassShardStride) : 0;
  v += dpp_i32<0x114, 0xf>(v, 0);
  v += dpp_i32<0x112, 0xf>(v, 0);
  v += dpp_i32<0x111, 0xf>(v, 0);
  return __builtin_amdgcn_readlane(v, kPassShards - 1) >= total;
}

// One wave waits for the group whose arrival shards are `gate` to be complete.  Bounded: gives up (and tells
// everybody) after timeout_ticks of the 100 MHz clock, or as soon as somebody else has.  The knobs are read here,
// on the slow path only (through a laundered pointer: the loads are not hoisted out of the item loop, where they
// would hold three more scalar registers for the whole kernel).
__device__ __forceinline__ void gate_wait(const PassArgs* ap, const unsigned* gate, int total, int item, int lane) {
  if (gate_is_open(gate, total, lane)) return;
  asm volatile("" : "+s"(ap));
  const PassArgs a = ld_const(ap);
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (unsigned spins = 1;; ++spins) {
    for (int i = 0; i < a.poll_sleep; ++i) __builtin_amdgcn_s_sleep(2);
    if (gate_is_open(gate, total, lane)) return;
    if ((spins & 15u) == 0) {
      const bool late = __builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)a.timeout_ticks;
      if (late || ld_agent_u32(a.status + kPassStatusError) != 0) {
        if (late && lane == 0) {
          // (diagnostics: what the gate looked like when the wait gave up)
          unsigned have = 0;
          for (int sh = 0; sh < kPassShards; ++sh) have += ld_agent_u32(gate + sh * kPassShardStride);
          __hip_atomic_store(SQLLM_GLOBAL(unsigned, a.status + 2), have, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(SQLLM_GLOBAL(unsigned, a.status + 3), (unsigned)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(SQLLM_GLOBAL(unsigned, a.status + 4), (unsigned)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          unsigned fresh = 0;  // the same shards through read-modify-write atomics, which execute at the memory side
          for (int sh = 0; sh < kPassShards; ++sh)
            fresh += __hip_atomic_fetch_add(SQLLM_GLOBAL(unsigned, const_cast<unsigned*>(gate) + sh * kPassShardStride), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(SQLLM_GLOBAL(unsigned, a.status + 5), fresh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          unsigned again = 0;
          for (int sh = 0; sh < kPassShards; ++sh) again += ld_agent_u32(gate + sh * kPassShardStride);
          __hip_atomic_store(SQLLM_GLOBAL(unsigned, a.status + 6), again, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(SQLLM_GLOBAL(unsigned, a.status + kPassStatusItem), (unsigned)item, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(SQLLM_GLOBAL(unsigned, a.status + kPassStatusError), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
      }
    }
  }
}

__device__ __forceinline__ void arrive(unsigned* shards, int item) {
  __hip_atomic_fetch_add(SQLLM_GLOBAL(unsigned, shards + (item % kPassShards) * kPassShardStride), 1u, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
}

// the sparse roles call this right before their first read of vec (sqllm_roles.h: GATE)
struct SparseGate {
  const PassArgs* ap;
  const unsigned* gate;
  int total, item;
  bool shut;
  __device__ __forceinline__ void operator()() const {
    if (!shut) return;  // (workgroup-uniform)
    if (threadIdx.x < 64) gate_wait(ap, gate, total, item, threadIdx.x);
    __syncthreads();
  }
};

#define SQLLM_PASS_HOT_OPERANDS(h) \
  "s"(h.q), "s"(h.y), "s"(h.lut), "s"(h.x), "s"(h.arrive), "s"(h.K), "s"(h.N), "s"(h.gate_group), "s"(h.gate_total)
#define SQLLM_PASS_ITEM_OPERANDS(i) "s"(i.seg_role), "s"(i.bid), "s"(i.u_beg), "s"(i.u_end)

template <int BITS> struct PassFmt;
template <> struct PassFmt<4> {
  static constexpr int kNbuf = 4;              // steps per chunk
  static constexpr int kNxr = 2;               // x registers per chunk (one serves two steps)
  static constexpr int kNe = 4;                // codebook values a thread stages
  static constexpr int kCodebookFloats = 2048;  // 4 column sub-tables: 2 pairs x 16 entries x 256 B
};
template <> struct PassFmt<3> {
  static constexpr int kNbuf = 1;
  static constexpr int kNxr = 2;
  static constexpr int kNe = 9;
  static constexpr int kCodebookFloats = 8192;  // 4 columns x 64 pair entries x 128 B
};

constexpr int kPassWaves = 8;
constexpr int kPassStep = kPassWaves * 4;  // units one workgroup step covers

template <int BITS>
constexpr int pass_lds_floats() {
  return cmax(PassFmt<BITS>::kCodebookFloats + kPassWaves * kTileN, cmax(2 * kCsrSpanMax, kTopxLds));
}

// Loads of a dense item that do not depend on vec: the tile's codebook values this thread stages, and the wave's
// first chunk of packed weights.  (Layouts: sqllm_kernels.hip, dense_role.)
template <int BITS>
__device__ __forceinline__ void dense_issue(const PassSegHot& sg, const PassItem& item, int lane, int wave,
                                            u32x4 (&w0)[PassFmt<BITS>::kNbuf][Fmt<BITS>::kRows], float (&ev)[PassFmt<BITS>::kNe]) {
  using F = Fmt<BITS>;
  constexpr int R = F::kRows, L = F::kLut, NBUF = PassFmt<BITS>::kNbuf;
  const int i16 = lane & 15, grp = lane >> 4;
  const int col0 = item.bid;
  const int u_last = item.u_end - 1;
  const int u_wave = item.u_beg + wave * 4;
  const int row_stride = sg.N / 4;
  int cidx = col0 / 4 + i16;
  if (cidx > row_stride - 1) cidx = row_stride - 1;
  const char* qbase = as_global(reinterpret_cast<const char*>(sg.q));
  const uint32_t lane_bytes = 16u * (uint32_t)cidx;
  const uint32_t row_bytes = 16u * (uint32_t)row_stride;
  const uint32_t unit_bytes = (uint32_t)R * row_bytes;
  {
    int c = col0 + 4 * i16 + (BITS == 3 ? grp : 2 * (wave & 1) + (lane >> 5));
    if (c > sg.N - 1) c = sg.N - 1;
    const float* src = as_global(sg.lut) + (size_t)c * L;
    if constexpr (BITS == 3) {
      const f32x4 ta = *reinterpret_cast<const f32x4*>(src), tb4 = *reinterpret_cast<const f32x4*>(src + 4);
      ev[0] = ta.x; ev[1] = ta.y; ev[2] = ta.z; ev[3] = ta.w;
      ev[4] = tb4.x; ev[5] = tb4.y; ev[6] = tb4.z; ev[7] = tb4.w;
      ev[8] = src[wave];
    } else {
      const f32x4 t = *reinterpret_cast<const f32x4*>(src + (wave >> 1) * 4);
      ev[0] = t.x; ev[1] = t.y; ev[2] = t.z; ev[3] = t.w;
    }
  }
#pragma unroll
  for (int s = 0; s < NBUF; ++s) {
    int uu = u_wave + grp + s * kPassStep;
    if (uu > u_last) uu = u_last;
    const uint32_t off = __umul24((uint32_t)uu, unit_bytes) + lane_bytes;
#pragma unroll
    for (int r = 0; r < R; ++r)
      w0[s][r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(qbase + (off + r * row_bytes)));
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// the persistent kernel
// ------------------------------------------------------------------------------------------------
template <int BITS>
__global__ void __launch_bounds__(kPassWaves * 64, 8) sqllm_pass_kernel(const PassArgs* ap) {
  using F = Fmt<BITS>;
  using P = PassFmt<BITS>;
  constexpr int R = F::kRows, NBUF = P::kNbuf, NXR = P::kNxr, NE = P::kNe;
  constexpr int T = kPassWaves * 64;
  constexpr int kLds = pass_lds_floats<BITS>();
  __shared__ __attribute__((aligned(16))) float lds[kLds + 4];  // + the epilogue ticket and the next-item slot, which no role overwrites
  unsigned* ticket = reinterpret_cast<unsigned*>(lds + kLds);
  if (threadIdx.x == 0) *ticket = 0u;
  const PassItem* const items = ld_const(&ap->items);
  const PassSeg* const segs = ld_const(&ap->segs);
  const int n_items = ld_const(&ap->n_items);
#ifdef SQLLM_ABLATION_BUILD
  unsigned long long* const tl = ld_const(&ap->timeline);
#define SQLLM_PASS_STAMP(I, COND) if (tl && (COND)) tl[4ull * (unsigned)it + (I)] = __builtin_amdgcn_s_memrealtime();
#else
#define SQLLM_PASS_STAMP(I, COND)
#endif

  // work items come from one queue in pass order: a ticket per item, taken one item ahead by thread 0 and handed
  // to the workgroup through LDS (slot kLds + 1) behind the barriers the item has anyway
  unsigned* const head = ld_const(&ap->status) + kPassStatusHead;
  unsigned* next_slot = reinterpret_cast<unsigned*>(lds + kLds + 1);
  if (threadIdx.x == 0) *next_slot = __hip_atomic_fetch_add(SQLLM_GLOBAL(unsigned, head), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  int it = __builtin_amdgcn_readfirstlane((int)*next_slot);
  if (it >= n_items) return;
  int open_upto = -1;  // highest group this workgroup has seen complete (a workgroup's items come in pass order)
  PassItem item = ld_const(items + it);
  PassSegHot sg = ld_const(&segs[item.seg_role & 0xffffff].hot);
  asm volatile("" ::SQLLM_PASS_HOT_OPERANDS(sg), SQLLM_PASS_ITEM_OPERANDS(item));

  for (;;) {
    // per-lane constants are re-derived per item from a laundered thread id: hoisted out of the loop they would be
    // live across every role (the kernel lives on fitting 64 VGPRs: four workgroups per CU)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, grp = lane >> 4;
    const int role = item.seg_role >> 24;
    const int seg_index = item.seg_role & 0xffffff;
    int nit;
    bool more;
    const bool shut = sg.gate_group > open_upto;
    unsigned* const my_arrive = sg.arrive;
    const unsigned* const gate = sg.arrive - kPassGroupStride;
    const int gate_total = sg.gate_total, gate_group = sg.gate_group;
    if (role == kPassDense) {
      // ---- geometry of this item (wave-uniform scalars + the lane's byte offsets), as in dense_role ----
      const int col0 = item.bid;
      const int u_end = item.u_end;
      const int u_last = u_end - 1;
      const int u_wave = item.u_beg + wave * 4;
      const int N = sg.N;
      const int row_stride = N / 4;
      int cidx = col0 / 4 + i16;
      if (cidx > row_stride - 1) cidx = row_stride - 1;
      const char* qbase = as_global(reinterpret_cast<const char*>(sg.q));
      const char* xbase = reinterpret_cast<const char*>(sg.x);
      float* const y = sg.y;
      const uint32_t lane_bytes = 16u * (uint32_t)cidx;
      const uint32_t row_bytes = 16u * (uint32_t)row_stride;
      const uint32_t unit_bytes = (uint32_t)R * row_bytes;

      __syncthreads();  // A: every wave is done with the previous item's codebooks, its combine has read the slabs

      // ---- run ahead of the gate: this item's codebook values and first chunk of weights.  (Issued HERE, not
      // beside the previous item's epilogue: the whole chip finishes a group at once, and an arrival that waits
      // behind 32 MB of everybody's prefetch delays every group by the time that burst takes -- measured: 28 us per
      // group instead of ~6, profiles/r04_pass_first_run.txt.) ----
      u32x4 w0[NBUF][R];
      float ev[NE];
      SQLLM_PASS_STAMP(0, threadIdx.x == 0)  // item begun: prefetch goes out
      dense_issue<BITS>(sg, item, lane, wave, w0, ev);

      // the ticket of the NEXT item goes out now; its round trip hides behind the codebook / gate wait
      unsigned ticket_next = 0;
      if (threadIdx.x == 0) ticket_next = __hip_atomic_fetch_add(SQLLM_GLOBAL(unsigned, head), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

      // ---- stage the codebooks (row-wise: sqllm_kernels.hip, dense_role) ----
      if constexpr (BITS == 3) {
        char* dst = reinterpret_cast<char*>(lds) + grp * 8192 + wave * 8 * 128 + i16 * 8;
#pragma unroll
        for (int i0 = 0; i0 < 8; ++i0) *reinterpret_cast<f32x2*>(dst + i0 * 128) = f32x2{ev[i0], ev[8]};
      } else {
        float* dst = lds + ((wave & 1) * 4096 + (wave >> 1) * 4 * 256) / 4 + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[i * 64] = ev[i];
      }
      // ---- the gate: vec is first consumed below ----
      if (shut && wave == 0) gate_wait(ap, gate, gate_total, it, lane);
      if (threadIdx.x == 0) *next_slot = ticket_next;
      __syncthreads();  // B: codebooks visible, gate open, next ticket in its slot
      SQLLM_PASS_STAMP(1, threadIdx.x == 0)  // gate passed
      if (shut) open_upto = gate_group;

      f32x2 acc[2][1] = {{f32x2{0.f, 0.f}}, {f32x2{0.f, 0.f}}};
      f32x2 accp[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
      const uint32_t lane_off = 4 * (i16 + 16 * (grp & 1));
      uint32_t tb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) tb[j] = j * 8192 + 8 * i16;

      auto load_x = [&](int u, float (&xs)[NXR][1]) {
        if constexpr (BITS == 4) {
#pragma unroll
          for (int s2 = 0; s2 < NBUF / 2; ++s2) {
            int uu = u + grp + (2 * s2 + (i16 >> 3)) * kPassStep;
            if (uu > u_last) uu = u_last;
            xs[s2][0] = ld_x<true>(reinterpret_cast<const float*>(xbase + 4u * (8u * (uint32_t)uu + (i16 & 7))));
          }
        } else {
#pragma unroll
          for (int s = 0; s < NBUF; ++s) {
            int uu = u + grp + s * kPassStep;
            if (uu > u_last) uu = u_last;
            const float* xp = reinterpret_cast<const float*>(xbase + 4u * (32u * (uint32_t)uu + i16));
            xs[2 * s][0] = ld_x<true>(xp);
            xs[2 * s + 1][0] = ld_x<true>(xp + 16);
          }
        }
      };
      auto load_w = [&](int u, u32x4 (&w)[NBUF][R]) {
#pragma unroll
        for (int s = 0; s < NBUF; ++s) {
          int uu = u + grp + s * kPassStep;
          if (uu > u_last) uu = u_last;
          const uint32_t off = __umul24((uint32_t)uu, unit_bytes) + lane_bytes;
#pragma unroll
          for (int r = 0; r < R; ++r)
            w[s][r] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(qbase + (off + r * row_bytes)));
        }
      };
      auto decode_chunk = [&](int u, const u32x4 (&w)[NBUF][R], const float (&xs)[NXR][1]) {
        if constexpr (BITS == 4) {
#pragma unroll
          for (int s2 = 0; s2 < NBUF / 2; ++s2) {
            const int ua = u + 2 * s2 * kPassStep, ub = ua + kPassStep;
            if (ua < u_end) step4_half<1, 0, 0>(w[2 * s2][0], xs[s2], ua + grp < u_end, lane_off, acc);
            if (ub < u_end) step4_half<1, 8, 0>(w[2 * s2 + 1][0], xs[s2], ub + grp < u_end, lane_off, acc);
          }
        } else {
#pragma unroll
          for (int s = 0; s < NBUF; ++s) {
            const int ua = u + s * kPassStep;
            if (ua < u_end) step3_pair(w[s], xs[2 * s][0], xs[2 * s + 1][0], ua + grp < u_end, tb, accp);
          }
        }
      };

      {
        float x0[NXR][1];
        load_x(u_wave, x0);
        // the next item's descriptor: fetched while the first x values are on their way
        nit = __builtin_amdgcn_readfirstlane((int)*next_slot);
        more = nit < n_items;
        if (more) {
          item = ld_const(items + nit);
          sg = ld_const(&segs[item.seg_role & 0xffffff].hot);
          asm volatile("" ::SQLLM_PASS_HOT_OPERANDS(sg), SQLLM_PASS_ITEM_OPERANDS(item));
        }
        __builtin_amdgcn_sched_barrier(0);
        decode_chunk(u_wave, w0, x0);
      }
      for (int u0 = u_wave + NBUF * kPassStep; u0 < u_end; u0 += NBUF * kPassStep) {  // scalar loop
        u32x4 w[NBUF][R];
        float xs[NXR][1];
        load_w(u0, w);
        load_x(u0, xs);
        __builtin_amdgcn_sched_barrier(0);
        decode_chunk(u0, w, xs);
      }
      if constexpr (BITS == 3) {
        acc[0][0] = f32x2{accp[0].x + accp[0].y, accp[1].x + accp[1].y};
        acc[1][0] = f32x2{accp[2].x + accp[2].y, accp[3].x + accp[3].y};
      }

      SQLLM_PASS_STAMP(2, threadIdx.x == 0)  // wave 0 done decoding
      // ---- epilogue, first half: fold the lane rows, park the wave's 64 partial sums in its slab ----
      float* red = lds + P::kCodebookFloats;  // [wave][64]
      {
        float col[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = (j & 1) ? acc[j >> 1][0].y : acc[j >> 1][0].x;
          v += __shfl_xor(v, 16, 64);
          v += __shfl_xor(v, 32, 64);
          col[j] = v;
        }
        if (grp == 0) *reinterpret_cast<f32x4*>(red + wave * kTileN + 4 * i16) = f32x4{col[0], col[1], col[2], col[3]};
      }
      // ---- epilogue, second half (barrier-free: slabs + ticket): the last wave sums, one atomic per column ----
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      unsigned t = 0;
      if (lane == 0) t = atomicAdd(ticket, 1u);
      t = __builtin_amdgcn_readfirstlane(t);
      if (t == kPassWaves - 1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int c = col0 + lane;
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < kPassWaves; ++w) sum += red[w * kTileN + lane];
        if (c < N) acc_add(y + c, sum);
        if (lane == 0) *ticket = 0u;
        // the item is complete once its accumulations have been acknowledged
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) arrive(my_arrive, it);
        SQLLM_PASS_STAMP(3, lane == 0)  // accumulations acknowledged, arrival issued
      }
    } else {
      // ---- sparse item: the role's own code (sqllm_roles.h), gated where it first reads vec ----
      const PassSegSparse sp = ld_const(&segs[seg_index].sp);
      const float* const x = as_global(sg.x);
      float* const y = as_global(sg.y);
      const int K = sg.K, N = sg.N, bid = item.bid;
      __syncthreads();  // the previous item is done with the LDS
      unsigned ticket_next = 0;
      if (threadIdx.x == 0) ticket_next = __hip_atomic_fetch_add(SQLLM_GLOBAL(unsigned, head), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const SparseGate g{ap, gate, gate_total, it, shut};
      if (role == kPassCsr)
        csr_role<T, 1, float, float, false, true, SparseGate>(x, y, as_global(sp.rows), as_global(sp.cols), as_global(sp.vals), sp.nnz, K, N, 0, 1, bid, lds, nullptr, 0,
                                                              nullptr, 0, nullptr, g);
      else
        topx_role<T, float, float, true, SparseGate>(x, y, as_global(sp.full_rows), as_global(sp.full_idx), sp.topX, K, N, 0, 1, bid, lds, g);
      if (shut) open_upto = gate_group;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its accumulations are acknowledged
      if (threadIdx.x == 0) *next_slot = ticket_next;
      __syncthreads();
      if (threadIdx.x == 0) arrive(my_arrive, it);
      nit = __builtin_amdgcn_readfirstlane((int)*next_slot);
      more = nit < n_items;
      if (more) {
        item = ld_const(items + nit);
        sg = ld_const(&segs[item.seg_role & 0xffffff].hot);
        asm volatile("" ::SQLLM_PASS_HOT_OPERANDS(sg), SQLLM_PASS_ITEM_OPERANDS(item));
      }
    }
    if (!more) break;
    it = nit;
  }
}

int pass_blocks_per_cu(int bits) {
  int n = 0;
  hipError_t e = bits == 4 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sqllm_pass_kernel<4>, kPassWaves * 64, 0)
                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sqllm_pass_kernel<3>, kPassWaves * 64, 0);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

// The launch's state words (status, ticket head, arrival shards) are zeroed by a kernel of the library's own: a
// hipMemsetAsync captured into a HIP graph wrote garbage into the region from the second replay on (ROCm 7.2,
// 6272-byte region: the bytes that came back were device addresses -- profiles/r04_pass_graph_memset.txt).
__global__ void __launch_bounds__(256) sqllm_pass_zero(unsigned* p, int words) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < words) p[i] = 0u;
}

hipError_t zero_pass_state(unsigned* state, int words, hipStream_t stream) {
  hipLaunchKernelGGL(sqllm_pass_zero, dim3((words + 255) / 256), dim3(256), 0, stream, state, words);
  return hipGetLastError();
}

hipError_t launch_pass(int bits, const PassArgs* device_args, int grid, hipStream_t stream, hipEvent_t e0, hipEvent_t e1) {
  if (grid < 1) return hipErrorInvalidValue;
  auto kern = bits == 4 ? sqllm_pass_kernel<4> : sqllm_pass_kernel<3>;
  if (e0 || e1) hipExtLaunchKernelGGL(kern, dim3(grid), dim3(kPassWaves * 64), 0, stream, e0, e1, 0, device_args);
  else hipLaunchKernelGGL(kern, dim3(grid), dim3(kPassWaves * 64), 0, stream, device_args);
  return hipGetLastError();
}

}  // namespace sqllm
