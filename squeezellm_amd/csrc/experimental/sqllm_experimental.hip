// sqllm_experimental.hip -- host side of the MEASUREMENT library (python -m squeezellm_amd.build --ablation ->
// libsqllm_hip_ablation.so; never part of the product library): the kernels that were built, measured on MI355X and
// NOT adopted -- the streaming batch-1 kernel (sqllm_stream.hip), the column-pair-table kernel (sqllm_pair.hip), the
// dependency-gated persistent pass (sqllm_pass.hip) -- their options, their routing, and the measurement knobs of the
// product kernels (ablation bits, LDS pad, timeline probes).  Everything here reaches the product's host layer
// through the hooks of sqllm_host.h; the product sources contain no trace of it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <vector>

#include "sqllm_hip.h"
#include "sqllm_host.h"
#include "sqllm_kernels.h"
#include "sqllm_pass.h"
#include "sqllm_pass_api.h"

#ifndef SQLLM_ABLATION_BUILD
#error "csrc/experimental/ belongs to the measurement library (python -m squeezellm_amd.build --ablation)"
#endif

using namespace sqllm_host;

namespace {

struct ExpKnobs {
  std::atomic<int> ablate{0};          // ablation bits of the fused kernel (sqllm_decode.h: ABL)
  std::atomic<int> lds_pad{0};         // unused dynamic LDS per workgroup, bytes (caps the workgroups per CU)
  std::atomic<int> ablate_csr{0};      // ablation bits of the CSR role
  std::atomic<int> pair4{-1};          // 4-bit batch-1 operator launches on the column-pair-table kernel: -1 default (off), 0 / 1
  std::atomic<int> pair4_min_mb{12};   // ... from this many MB of packed weights per launch
  std::atomic<int> stream{-1};         // batch-1 operator launches on the streaming kernel: -1 = default (off), 0 / 1
  std::atomic<void*> timeline{nullptr};  // per-workgroup (or, for the pass, per-item) timestamp buffer
  std::atomic<int> pass_poll_sleep{4};     // gated pass: s_sleep(2) units between two polls of a gate
  std::atomic<int> pass_timeout_ms{2000};  // ... a gate that stays shut this long ends the launch with status 1
  std::atomic<int> pass_wgs_per_cu{0};     // ... workgroups per CU the kernel is launched with (0 = the occupancy query's answer)
  std::atomic<int> skip_prepare_small{0};  // TIMING ONLY: the fused small launch without sqllm_prepare_small / sqllm_transpose_small in front (garbage results)
};
ExpKnobs g_xknobs[kMaxDevices];
ExpKnobs& xknobs() { return g_xknobs[device_slot()]; }

// Column-pair-table kernel (sqllm_pair.hip): 4-bit operator launches at batch 1 whose packed weights are large
// enough to pay for the 64 KiB tables (option pair4_min_mb, MB per launch).
bool takes_pair4_path(const sqllm_op* ops, int n) {
  const int v = xknobs().pair4.load(std::memory_order_relaxed);
  if (v == 0 || ops[0].bits != 4) return false;
  if (v < 0) return false;  // default: off until measured
  double mb = 0.0;
  for (int i = 0; i < n; ++i) {
    if (ops[i].batch > 1 || (ops[i].N % 4) != 0) return false;
    mb += (double)ops[i].K * ops[i].N / 2.0 / 1e6;
  }
  return mb >= (double)xknobs().pair4_min_mb.load(std::memory_order_relaxed);
}

// Streaming batch-1 kernel (sqllm_stream.hip): does this launch take it, and with what geometry?
bool takes_stream_path(const sqllm_op* ops, int n) {
  const int v = xknobs().stream.load(std::memory_order_relaxed);
  if (v <= 0 || ops[0].bits != 4) return false;  // measurement library only, 4-bit only, off unless asked for
  const int kK = ops[0].bits == 4 ? 8 : 32;
  const uint32_t S = (uint32_t)((ops[0].K / kK + 3) / 4);
  uint64_t tiles = 0;
  for (int i = 0; i < n; ++i) {
    if (ops[i].batch > 1) return false;
    // dead lanes / steps are pushed out of range by adding 2^31 to their offsets: operands stay below that
    if ((uint64_t)ops[i].K / 32u * (uint64_t)ops[i].bits * (uint64_t)ops[i].N * 4u >= (1ull << 31)) return false;
    if ((uint64_t)ops[i].K * 4u >= (1ull << 31)) return false;
    tiles += (uint64_t)(ops[i].N + sqllm::kTileN - 1) / sqllm::kTileN;
  }
  // tile of a step by multiplication with m = ceil(2^32 / S): exact while step * (m * S - 2^32) < 2^32
  const uint64_t m = ((1ull << 32) + S - 1) / S;
  if (m >= (1ull << 32)) return false;  // S == 1: every step is a tile, no division needed -- rare, use the fused kernel
  if (tiles * S * (m * S - (1ull << 32)) >= (1ull << 32)) return false;
  return true;
}

// The dense work of the launch = the ops' 64-column tiles back to back, `steps_per_tile` steps each (a step
// = 4 units = one wave load); equal contiguous ranges, one per workgroup, ONE resident round: as many
// workgroups as the chip holds at once minus the launch's sparse-role workgroups (they come first in the
// grid and hold slots of their own), at least two steps per wave where the launch is small.  A range
// may touch at most `pieces` tiles (their codebooks are all staged up front).
void make_plan_stream(const sqllm_op* ops, int n, int sparse_blocks, sqllm::StreamArgs* sa) {
  const int bits = ops[0].bits;
  const int kK = bits == 4 ? 8 : 32;
  const int pieces = bits == 4 ? sqllm::kStreamPieces4 : sqllm::kStreamPieces3;
#ifdef SQLLM_STREAM_WGCU
  const int wg_per_cu = SQLLM_STREAM_WGCU;
#else
  const int wg_per_cu = bits == 4 ? 4 : 2;
#endif
  memset(sa, 0, sizeof(*sa));
  sa->x = static_cast<const float*>(ops[0].vec);
  sa->K = ops[0].K;
  sa->units_total = ops[0].K / kK;
  sa->steps_per_tile = (sa->units_total + 3) / 4;
  sa->s_magic = (uint32_t)(((1ull << 32) + sa->steps_per_tile - 1) / sa->steps_per_tile);
  sa->n_seg = n;
  int tiles = 0;
  for (int i = 0; i < sqllm::kMaxSegments; ++i) {
    sqllm::StreamSeg& sg = sa->seg[i];
    if (i < n) {
      sg.q = reinterpret_cast<const uint32_t*>(ops[i].qweight);
      sg.y = ops[i].mul;
      sg.lut = ops[i].lookup_table;
      sg.N = ops[i].N;
      sg.tile0 = tiles;
      tiles += (ops[i].N + sqllm::kTileN - 1) / sqllm::kTileN;
    } else {
      sg = sa->seg[0];
      sg.tile0 = 0x7fffffff;
    }
  }
  const long long total = (long long)tiles * sa->steps_per_tile;
  sa->total_steps = (int)total;
  int target = knobs().target_wgs.load(std::memory_order_relaxed);
  if (target <= 0) {
    const int slots = wg_per_cu * cu_count();
    target = slots - sparse_blocks;
    if (target < cu_count()) target = cu_count();
    const long long by_work = total / (2 * sqllm::kWaves);  // >= 2 steps per wave
    if (target > by_work) target = (int)(by_work < 1 ? 1 : by_work);
  }
  long long upw = (total + target - 1) / target;
  if (upw < 1) upw = 1;
  // at most `pieces` tiles per range: a range of upw steps touches <= ceil(upw / S) + 1 tiles
  const long long upw_max = (long long)(pieces - 1) * sa->steps_per_tile;
  if (upw > upw_max) upw = upw_max;
  sa->steps_per_wg = (int)upw;
  sa->n_dense = (int)((total + upw - 1) / upw);
}


// ---- routing of batch-1 operator launches to the experimental kernels ----
int route_stream(const sqllm_op* ops, int n, sqllm_stream_t stream, hipEvent_t e0, hipEvent_t e1) {
    // streaming kernel: [sparse-role workgroups of every op | pad to 8 | dense ranges]
    sqllm::GroupArgs ga;
    memset(&ga, 0, sizeof(ga));
    ga.n_seg = n;
    int block = 0;
    for (int i = 0; i < n; ++i) {
      const sqllm_op* op = &ops[i];
      int rc = validate(op);
      if (rc == SQLLM_OK) rc = validate_csr_values(op, stream);
      if (rc != SQLLM_OK) return rc;
      if (op->vec != ops[0].vec || op->K != ops[0].K || op->bits != ops[0].bits) return SQLLM_E_GROUP;
      sqllm::Segment& sg = ga.seg[i];
      fill_segment(op, &sg);
      make_plan(op, &sg.gm, n);
      sg.gm.dense_blocks = 0;
      sg.gm.dense_block0 = sg.gm.csr_blocks + sg.gm.topx_blocks;
      ga.block0[i] = block;
      block += sg.gm.csr_blocks + sg.gm.topx_blocks;
    }
    for (int i = n; i <= sqllm::kMaxSegments; ++i) ga.block0[i] = block;
    sqllm::StreamArgs sa;
    make_plan_stream(ops, n, block, &sa);
    sa.dense_block0 = (block + 7) / 8 * 8;
    sa.probe = static_cast<unsigned long long*>(xknobs().timeline.load(std::memory_order_relaxed));
    return static_cast<int>(sqllm::launch_stream(ops[0].bits, sa, ga, static_cast<hipStream_t>(stream), e0, e1,
                                                 xknobs().ablate.load(std::memory_order_relaxed)));
  }

int route_pair4(const sqllm_op* ops, int n, sqllm_stream_t stream, hipEvent_t e0, hipEvent_t e1) {
  if (n < 1 || n > sqllm::kMaxSegments) return SQLLM_E_GROUP;
  sqllm::LaunchArgs a;
  a.ev_start = e0;
  a.ev_stop = e1;
  a.x = ops[0].vec;
  a.ga.n_seg = n;
  int block = 0;
  for (int i = 0; i < n; ++i) {
    const sqllm_op* op = &ops[i];
    int rc = validate(op);
    if (rc == SQLLM_OK) rc = validate_csr_values(op, stream);
    if (rc != SQLLM_OK) return rc;
    if (op->vec != ops[0].vec || op->K != ops[0].K || op->bits != ops[0].bits) return SQLLM_E_GROUP;
    sqllm::Segment& sg = a.ga.seg[i];
    fill_segment(op, &sg);
    make_plan(op, &sg.gm, n, sqllm::kMaxSlices, 16);  // 16-wave workgroups
    sg.gm.sparse_last = 0;
    a.ga.block0[i] = block;
    block += (sg.gm.dense_block0 + sg.gm.dense_blocks + 7) / 8 * 8;
  }
  for (int i = n; i <= sqllm::kMaxSegments; ++i) a.ga.block0[i] = block;
  for (int i = n; i < sqllm::kMaxSegments; ++i) memset(&a.ga.seg[i], 0, sizeof(sqllm::Segment));
  return static_cast<int>(sqllm::launch_pair4(a, static_cast<hipStream_t>(stream)));
}

bool route(const sqllm_op* ops, int n, sqllm_stream_t stream, hipEvent_t e0, hipEvent_t e1, int* rc) {
  if (!ops || n < 1) return false;
  for (int i = 0; i < n; ++i)
    if (ops[i].batch > 1) return false;  // (the batched routes of the product come first anyway)
  if (takes_stream_path(ops, n)) { *rc = route_stream(ops, n, stream, e0, e1); return true; }
  if (takes_pair4_path(ops, n)) { *rc = route_pair4(ops, n, stream, e0, e1); return true; }
  return false;
}

void decorate(sqllm::LaunchArgs* a) {
  a->ablate = xknobs().ablate.load(std::memory_order_relaxed);
  a->lds_pad = xknobs().lds_pad.load(std::memory_order_relaxed);
  if (!a->linear)  // operator launches do not use Segment::bias: it carries the timeline buffer (tools/timeline.py)
    for (int i = 0; i < a->ga.n_seg; ++i) a->ga.seg[i].bias = static_cast<const float*>(xknobs().timeline.load(std::memory_order_relaxed));
}

int csr_ablation_bits() { return xknobs().ablate_csr.load(std::memory_order_relaxed); }
bool skip_prepare_small() { return xknobs().skip_prepare_small.load(std::memory_order_relaxed) != 0; }

int set_option(const char* name, int value) {
  if (!strcmp(name, "stream")) { xknobs().stream.store(value > 1 ? -1 : value); return SQLLM_OK; }  // 0 off, 1 on, 2 default (off)
  if (!strcmp(name, "pair4")) { xknobs().pair4.store(value > 1 ? -1 : value); return SQLLM_OK; }    // 0 off, 1 on, 2 default (off)
  if (!strcmp(name, "pair4_min_mb")) { xknobs().pair4_min_mb.store(value); return SQLLM_OK; }
  if (!strcmp(name, "ablate")) { xknobs().ablate.store(value); return SQLLM_OK; }
  if (!strcmp(name, "lds_pad")) { xknobs().lds_pad.store(value); return SQLLM_OK; }
  if (!strcmp(name, "ablate_csr")) { xknobs().ablate_csr.store(value); return SQLLM_OK; }
  if (!strcmp(name, "pass_poll_sleep")) { xknobs().pass_poll_sleep.store(value); return SQLLM_OK; }
  if (!strcmp(name, "pass_timeout_ms")) { xknobs().pass_timeout_ms.store(value > 0 ? value : 1); return SQLLM_OK; }
  if (!strcmp(name, "pass_wgs_per_cu")) { xknobs().pass_wgs_per_cu.store(value); return SQLLM_OK; }
  if (!strcmp(name, "skip_prepare_small")) { xknobs().skip_prepare_small.store(value ? 1 : 0); return SQLLM_OK; }
  return SQLLM_E_OPTION;
}

int get_option(const char* name, int* value) {
  if (!strcmp(name, "stream")) { const int v = xknobs().stream.load(); *value = v < 0 ? 2 : v; return SQLLM_OK; }
  if (!strcmp(name, "pair4")) { const int v = xknobs().pair4.load(); *value = v < 0 ? 2 : v; return SQLLM_OK; }
  if (!strcmp(name, "pair4_min_mb")) { *value = xknobs().pair4_min_mb.load(); return SQLLM_OK; }
  if (!strcmp(name, "ablate")) { *value = xknobs().ablate.load(); return SQLLM_OK; }
  if (!strcmp(name, "lds_pad")) { *value = xknobs().lds_pad.load(); return SQLLM_OK; }
  if (!strcmp(name, "ablate_csr")) { *value = xknobs().ablate_csr.load(); return SQLLM_OK; }
  if (!strcmp(name, "pass_poll_sleep")) { *value = xknobs().pass_poll_sleep.load(); return SQLLM_OK; }
  if (!strcmp(name, "pass_timeout_ms")) { *value = xknobs().pass_timeout_ms.load(); return SQLLM_OK; }
  if (!strcmp(name, "pass_wgs_per_cu")) { *value = xknobs().pass_wgs_per_cu.load(); return SQLLM_OK; }
  if (!strcmp(name, "skip_prepare_small")) { *value = xknobs().skip_prepare_small.load(); return SQLLM_OK; }
  return SQLLM_E_OPTION;
}

struct InstallHooks {
  InstallHooks() {
    g_experimental.set_option = set_option;
    g_experimental.get_option = get_option;
    g_experimental.route = route;
    g_experimental.decorate = decorate;
    g_experimental.csr_ablation_bits = csr_ablation_bits;
    g_experimental.skip_prepare_small = skip_prepare_small;
  }
} g_install_hooks;

// ---- dependency-gated pass: workspace layout ----
// Workspace image:  [status words | arrival shards, kPassGroupStride dwords per group]  <- zeroed before every launch
//                   [PassArgs, 128 bytes] [PassSeg per op, 128-byte aligned] [PassItem per work item]

struct PassLayout {
  int64_t state_bytes, segs_offset, items_offset, total_bytes;
  int32_t n_ops, n_items;
};

int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

// shapes only: counts the work items and lays the workspace out (no pointer is dereferenced or stored)
int pass_layout(const sqllm_op* ops, const int32_t* group_sizes, int32_t n_groups, PassLayout* out) {
  if (n_groups < 1 || !ops || !group_sizes) return SQLLM_E_NULL;
  int64_t n_ops = 0, n_items = 0;
  for (int32_t g = 0; g < n_groups; ++g) {
    const int n = group_sizes[g];
    if (n < 1 || n > SQLLM_PASS_MAX_GROUP_OPS) return SQLLM_E_GROUP;
    for (int i = 0; i < n; ++i) {
      const sqllm_op* op = &ops[n_ops + i];
      if (op->bits != 3 && op->bits != 4) return SQLLM_E_BITS;
      if (op->K <= 0 || op->N <= 0 || (op->K % 32) != 0 || (op->N % 4) != 0) return SQLLM_E_SHAPE;
      if (op->batch > 1 || op->batch < 0) return SQLLM_E_BATCH;
      sqllm::KernelGeom gm;
      make_plan(op, &gm, n);
      n_items += (int64_t)gm.dense_blocks + gm.csr_blocks + gm.topx_blocks;
    }
    n_ops += n;
  }
  if (n_ops >= (1 << 24) || n_items > 0x7fffffff) return SQLLM_E_SHAPE;
  out->n_ops = (int32_t)n_ops;
  out->n_items = (int32_t)n_items;
  out->state_bytes = align_up(4ll * (sqllm::kPassStatusWords + (int64_t)n_groups * sqllm::kPassGroupStride), 128);
  out->segs_offset = out->state_bytes + 128;  // (the kernel's argument block sits in between)
  out->items_offset = align_up(out->segs_offset + (int64_t)sizeof(sqllm::PassSeg) * n_ops, 128);
  out->total_bytes = align_up(out->items_offset + (int64_t)sizeof(sqllm::PassItem) * n_items, 128);
  return SQLLM_OK;
}


}  // namespace

extern "C" {

// measurement build only (not in the header): device buffer of 8 x u64 per workgroup of the next launches
void sqllm_debug_set_timeline(void* buf) { xknobs().timeline.store(buf); }
// measurement build only: the streaming kernel's plan for a group of ops -> {takes_stream, n_dense, steps_per_wg, steps_per_tile, total_steps}
void sqllm_debug_stream_plan(const sqllm_op* ops, int n, int sparse_blocks, int* out) {
  sqllm::StreamArgs sa;
  out[0] = takes_stream_path(ops, n) ? 1 : 0;
  make_plan_stream(ops, n, sparse_blocks, &sa);
  out[1] = sa.n_dense; out[2] = sa.steps_per_wg; out[3] = sa.steps_per_tile; out[4] = sa.total_steps;
}


int64_t sqllm_pass_workspace_bytes(const sqllm_op* ops, const int32_t* group_sizes, int32_t n_groups) {
  PassLayout lay;
  const int rc = pass_layout(ops, group_sizes, n_groups, &lay);
  return rc == SQLLM_OK ? lay.total_bytes : (int64_t)rc;
}

int sqllm_pass_plan(const sqllm_op* ops, const int32_t* group_sizes, int32_t n_groups, void* workspace,
                    int64_t workspace_bytes, void* host_image, sqllm_pass* pass) {
  if (!pass || !host_image || !workspace) return SQLLM_E_NULL;
  PassLayout lay;
  int rc = pass_layout(ops, group_sizes, n_groups, &lay);
  if (rc != SQLLM_OK) return rc;
  if (workspace_bytes < lay.total_bytes || (reinterpret_cast<uintptr_t>(workspace) & 127u) != 0) return SQLLM_E_WORKSPACE;
  char* img = static_cast<char*>(host_image);
  memset(img, 0, (size_t)lay.total_bytes);
  char* dev = static_cast<char*>(workspace);
  auto* segs = reinterpret_cast<sqllm::PassSeg*>(img + lay.segs_offset);
  auto* items = reinterpret_cast<sqllm::PassItem*>(img + lay.items_offset);
  auto arrive_of = [&](int g) {
    return reinterpret_cast<unsigned*>(dev) + sqllm::kPassStatusWords + (size_t)g * sqllm::kPassGroupStride;
  };
  const int bits = ops[0].bits;
  int op0 = 0, n_item = 0, prev_total = 0;
  for (int32_t g = 0; g < n_groups; ++g) {
    const int n = group_sizes[g];
    const int first_item = n_item;
    sqllm::KernelGeom gm[SQLLM_PASS_MAX_GROUP_OPS];
    for (int i = 0; i < n; ++i) {
      const sqllm_op* op = &ops[op0 + i];
      rc = validate(op);
      if (rc != SQLLM_OK) return rc;
      if (op->bits != bits) return SQLLM_E_GROUP;  // one kernel instantiation serves the whole pass
      if (op->vec != ops[op0].vec || op->K != ops[op0].K) return SQLLM_E_GROUP;
      make_plan(op, &gm[i], n);
      sqllm::PassSeg& sg = segs[op0 + i];
      sg.hot.q = reinterpret_cast<const uint32_t*>(op->qweight);
      sg.hot.y = op->mul;
      sg.hot.lut = op->lookup_table;
      sg.hot.x = op->vec;
      sg.hot.arrive = arrive_of(g);
      // group g + 1 reads what group g wrote: its vec is gated on the completion of group g (the first group's vec
      // is complete before the launch, by stream order)
      sg.hot.gate_group = g > 0 ? g - 1 : -1;
      sg.hot.gate_total = prev_total;
      sg.hot.K = op->K;
      sg.hot.N = op->N;
      sg.sp.rows = gm[i].csr_blocks ? op->rows : nullptr;
      sg.sp.cols = gm[i].csr_blocks ? op->cols : nullptr;
      sg.sp.vals = gm[i].csr_blocks ? op->vals : nullptr;
      sg.sp.full_rows = gm[i].topx_blocks ? op->full_rows : nullptr;
      sg.sp.full_idx = gm[i].topx_blocks ? op->full_row_indices : nullptr;
      sg.sp.nnz = gm[i].nnz;
      sg.sp.topX = gm[i].topX;
      sg.sp.col_tiles = gm[i].col_tiles;
      sg.sp.units_total = gm[i].units_total;
      sg.sp.units_per_wg = gm[i].units_per_wg;
      sg.sp.group = g;
    }
    // a group's items in the order they are dealt: the latency-bound sparse items first (CSR chunks, top-X slabs),
    // then the dense tiles, K slice by K slice (consecutive workgroups = consecutive column tiles, as in the
    // one-launch-per-group kernel)
    for (int i = 0; i < n; ++i)
      for (int b = 0; b < gm[i].csr_blocks; ++b) items[n_item++] = {(op0 + i) | (sqllm::kPassCsr << 24), b, 0, 0};
    for (int i = 0; i < n; ++i)
      for (int b = 0; b < gm[i].topx_blocks; ++b) items[n_item++] = {(op0 + i) | (sqllm::kPassTopx << 24), b, 0, 0};
    for (int i = 0; i < n; ++i)
      for (int ks = 0; ks < gm[i].k_slices; ++ks) {
        const int u_beg = ks * gm[i].units_per_wg;
        const int u_end = u_beg + gm[i].units_per_wg < gm[i].units_total ? u_beg + gm[i].units_per_wg : gm[i].units_total;
        for (int ct = 0; ct < gm[i].col_tiles; ++ct)
          items[n_item++] = {(op0 + i) | (sqllm::kPassDense << 24), ct * sqllm::kTileN, u_beg, u_end};
      }
    prev_total = n_item - first_item;
    op0 += n;
  }
  if (n_item != lay.n_items) return SQLLM_E_SHAPE;  // (cannot happen: pass_layout counted with the same plans)
  {
    sqllm::PassArgs* a = reinterpret_cast<sqllm::PassArgs*>(img + lay.state_bytes);
    a->items = reinterpret_cast<const sqllm::PassItem*>(dev + lay.items_offset);
    a->segs = reinterpret_cast<const sqllm::PassSeg*>(dev + lay.segs_offset);
    a->status = reinterpret_cast<unsigned*>(dev);
    a->n_items = lay.n_items;
    a->poll_sleep = xknobs().pass_poll_sleep.load(std::memory_order_relaxed);
    const long long ticks = (long long)xknobs().pass_timeout_ms.load(std::memory_order_relaxed) * 100000ll;  // 100 MHz
    a->timeout_ticks = ticks > 0xffffffffll ? 0xffffffffu : (unsigned)ticks;
    a->timeline = static_cast<unsigned long long*>(xknobs().timeline.load(std::memory_order_relaxed));
  }
  memset(pass, 0, sizeof(*pass));
  pass->workspace = workspace;
  pass->workspace_bytes = lay.total_bytes;
  pass->segs_offset = lay.segs_offset;
  pass->items_offset = lay.items_offset;
  pass->state_bytes = (int32_t)lay.state_bytes;
  pass->bits = bits;
  pass->n_groups = n_groups;
  pass->n_ops = lay.n_ops;
  pass->n_items = lay.n_items;
  int per_cu = xknobs().pass_wgs_per_cu.load(std::memory_order_relaxed);
  if (per_cu <= 0) {
    per_cu = sqllm::pass_blocks_per_cu(bits);
    if (per_cu <= 0) per_cu = 4;  // (no device to ask: the kernel is built for four per CU -- tests/test_codegen_cpu.py)
  }
  // (work items are taken from a queue: the grid only has to be what the chip CAN hold, not what it WILL)
  long long grid = (long long)per_cu * cu_count();
  if (grid > lay.n_items) grid = lay.n_items;
  pass->grid = (int32_t)grid;
  pass->poll_sleep = xknobs().pass_poll_sleep.load(std::memory_order_relaxed);
  pass->timeout_ms = xknobs().pass_timeout_ms.load(std::memory_order_relaxed);
  return SQLLM_OK;
}

int sqllm_pass_build(const sqllm_op* ops, const int32_t* group_sizes, int32_t n_groups, void* workspace,
                     int64_t workspace_bytes, sqllm_pass* pass) {
  const int64_t need = sqllm_pass_workspace_bytes(ops, group_sizes, n_groups);
  if (need < 0) return (int)need;
  if (workspace_bytes < need) return SQLLM_E_WORKSPACE;
  std::vector<char> img((size_t)need);
  int rc = sqllm_pass_plan(ops, group_sizes, n_groups, workspace, workspace_bytes, img.data(), pass);
  if (rc != SQLLM_OK) return rc;
  for (int i = 0; i < pass->n_ops; ++i) {
    rc = validate_csr_values(&ops[i], nullptr);
    if (rc != SQLLM_OK) return rc;
  }
  const hipError_t e = hipMemcpy(workspace, img.data(), (size_t)need, hipMemcpyHostToDevice);
  return e == hipSuccess ? SQLLM_OK : (int)e;
}

static int pass_launch_with_events(const sqllm_pass* pass, sqllm_stream_t stream, hipEvent_t e0, hipEvent_t e1) {
  if (!pass || !pass->workspace) return SQLLM_E_NULL;
  if (pass->n_items < 1 || pass->grid < 1 || pass->state_bytes < 4 * sqllm::kPassStatusWords || pass->segs_offset != pass->state_bytes + 128 ||
      pass->items_offset + (int64_t)sizeof(sqllm::PassItem) * pass->n_items > pass->workspace_bytes)
    return SQLLM_E_WORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  char* ws = static_cast<char*>(pass->workspace);
  hipError_t e = sqllm::zero_pass_state(reinterpret_cast<unsigned*>(ws), pass->state_bytes / 4, s);
  if (e != hipSuccess) return (int)e;
  return (int)sqllm::launch_pass(pass->bits, reinterpret_cast<const sqllm::PassArgs*>(ws + pass->state_bytes), pass->grid, s, e0, e1);
}

int sqllm_pass_launch(const sqllm_pass* pass, sqllm_stream_t stream) { return pass_launch_with_events(pass, stream, nullptr, nullptr); }

int sqllm_pass_status(const sqllm_pass* pass, sqllm_stream_t stream, int32_t* error, int32_t* item) {
  if (!pass || !pass->workspace) return SQLLM_E_NULL;
  hipError_t e = hipStreamSynchronize(static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return (int)e;
  unsigned st[2] = {0, 0};
  e = hipMemcpy(st, pass->workspace, sizeof(st), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return (int)e;
  if (error) *error = (int32_t)st[sqllm::kPassStatusError];
  if (item) *item = (int32_t)st[sqllm::kPassStatusItem];
  return SQLLM_OK;
}

int sqllm_pass_profile(const sqllm_pass* pass, sqllm_stream_t stream, int32_t reps, float* avg_us) {
  if (!pass || !avg_us || reps < 1) return SQLLM_E_NULL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
    if (e0) (void)hipEventDestroy(e0);
    return (int)hipGetLastError();
  }
  int rc = SQLLM_OK;
  double sum = 0.0;
  for (int r = 0; r < reps && rc == SQLLM_OK; ++r) {
    rc = pass_launch_with_events(pass, stream, e0, e1);
    if (rc != SQLLM_OK) break;
    hipError_t e = hipStreamSynchronize(s);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e != hipSuccess) { rc = (int)e; break; }
    sum += ms * 1000.0;
  }
  *avg_us = (float)(sum / reps);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return rc;
}


}  // extern "C"
