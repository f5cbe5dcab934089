// sqllm_capi.hip -- the extern "C" surface declared in include/sqllm_hip.h: argument validation,
// launch planning, and the reference operator names as thin adapters over sqllm_launch().
// Replaces the reference's pybind11 layer squeezellm/quant_cuda.cpp:112-270 and the grid math of
// its launchers squeezellm/quant_cuda_kernel.cu:132-738 (no torch types cross this boundary).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <atomic>

#include "sqllm_hip.h"
#include "sqllm_host.h"
#include "sqllm_kernels.h"

namespace sqllm_host {

ExperimentalHooks g_experimental;
static Knobs g_knobs[kMaxDevices];

int device_slot() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();  // no usable device: not an error of the call being served
    dev = 0;
  }
  if (dev < 0 || dev >= kMaxDevices) dev = 0;
  return dev;
}

Knobs& knobs() { return g_knobs[device_slot()]; }

// The wide-batch paths take stream-ordered scratch (hipMallocAsync) per group of ops.  The default pool's
// release threshold is 0: every synchronisation hands the block back to the OS and the next call pays
// a real allocation + map.  Raise it once per device (never lower it) so that the pool keeps what one
// group needs (2048 rows x K = 22016: transposed vec 180 MB + bf16 planes 271 MB + slabs <= 32 MB).
static void keep_scratch_in_pool() {
  static std::atomic<unsigned> done{0};
  if (!knobs().scratch_pool_threshold.load(std::memory_order_relaxed)) return;  // opted out: the pool is left as the application set it
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) { (void)hipGetLastError(); return; }
  const unsigned bit = 1u << dev;
  if (done.load(std::memory_order_relaxed) & bit) return;
  hipMemPool_t pool = nullptr;
  if (hipDeviceGetDefaultMemPool(&pool, dev) != hipSuccess || !pool) { (void)hipGetLastError(); return; }  // (retried on the next call)
  uint64_t cur = 0;
  const uint64_t want = 1024ull << 20;
  if (hipMemPoolGetAttribute(pool, hipMemPoolAttrReleaseThreshold, &cur) != hipSuccess) { (void)hipGetLastError(); cur = 0; }
  if (cur < want) {
    uint64_t v = want;
    if (hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &v) != hipSuccess) { (void)hipGetLastError(); return; }  // (retried)
  }
  done.fetch_or(bit, std::memory_order_relaxed);  // only once it has succeeded
}

int cu_count() {
  int c = knobs().cu_count.load(std::memory_order_relaxed);
  if (c > 0) return c;
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
      prop.multiProcessorCount > 0)
    c = prop.multiProcessorCount;
  else
    c = 256;  // MI355X
  knobs().cu_count.store(c, std::memory_order_relaxed);
  return c;
}

int validate(const sqllm_op* op) {
  if (!op) return SQLLM_E_NULL;
  if (op->bits != 3 && op->bits != 4) return SQLLM_E_BITS;
  if (op->K <= 0 || op->N <= 0 || (op->K % 32) != 0 || (op->N % 4) != 0) return SQLLM_E_SHAPE;
  // the kernels address qweight with 32-bit byte offsets
  if ((uint64_t)op->K / 32u * (uint64_t)op->bits * (uint64_t)op->N * 4u >= (1ull << 32)) return SQLLM_E_SHAPE;
  if (op->batch < 0) return SQLLM_E_BATCH;
  if (!op->vec || !op->qweight || !op->mul || !op->lookup_table) return SQLLM_E_NULL;
  if ((reinterpret_cast<uintptr_t>(op->qweight) & 15u) != 0 ||
      (reinterpret_cast<uintptr_t>(op->lookup_table) & 15u) != 0)
    return SQLLM_E_ALIGN;
  if (op->rows) {
    if (op->nnz < 0) return SQLLM_E_SPARSE;
    if (op->nnz > 0 && (!op->cols || !op->vals)) return SQLLM_E_NULL;
  }
  if (op->full_rows) {
    if (op->topX < 0) return SQLLM_E_SPARSE;
    if (op->topX > 0 && !op->full_row_indices) return SQLLM_E_NULL;
  }
  return SQLLM_OK;
}

// option "validate_csr": a value check of rows[] on the device (blocks the host; debugging aid)
int validate_csr_values(const sqllm_op* op, sqllm_stream_t stream) {
  if (!knobs().validate_csr.load(std::memory_order_relaxed) || !op->rows || op->nnz <= 0) return SQLLM_OK;
  int bad = 0;
  hipError_t e = sqllm::check_csr(op->rows, op->N, op->nnz, static_cast<hipStream_t>(stream), &bad);
  if (e != hipSuccess) return static_cast<int>(e);
  return bad ? SQLLM_E_SPARSE : SQLLM_OK;
}

// Launch geometry.  The dense part is cut into 64-column tiles x K slices so that about `target`
// workgroups exist (1-3 per CU, 8 waves each, of the 4 that fit: the 7B shapes hold only
// ~32-90 KiB of weights per CU, so the grid must be wide rather than deep).  A slice is a whole
// number of workgroup steps (waves x 4 units) so only the last slice has a ragged end.
void make_plan(const sqllm_op* op, sqllm::KernelGeom* gm, int ops_in_launch, int max_slices, int waves) {
  const int kK = (op->bits == 4) ? 8 : 32;
  memset(gm, 0, sizeof(*gm));
  gm->K = op->K;
  gm->N = op->N;
  gm->batch = op->batch <= 0 ? 1 : op->batch;
  gm->col_tiles = (op->N + sqllm::kTileN - 1) / sqllm::kTileN;
  gm->units_total = op->K / kK;
  const int step = waves * 4;  // units one workgroup step covers
  int upw = knobs().groups_per_wave.load(std::memory_order_relaxed) * waves;
  if (upw <= 0) {
    int target = knobs().target_wgs.load(std::memory_order_relaxed);
    if (target <= 0) {
      // measured on MI355X (tools/sweep.py, bench.py): an op under ~12 MB of packed weights runs
      // best with one 8-wave workgroup per CU when it shares its launch with others (q/k/v) and
      // with two when it is alone (o_proj); larger ones with ~3 per CU (gate/up) or 4 when alone
      // (down_proj) -- the chip holds four per CU.  In between (the 13B q/k/v at 4 bits, 13 MB each, sharing a
      // launch): two per CU and op -- three made 2400 workgroups of the group, 6 % slower
      // (profiles/r03_target_wgs_groups_batch1.txt)
      const double mb = (double)op->K * op->N * op->bits / 8.0 / 1e6;
      const bool alone = ops_in_launch <= 1;
      // ... and the DENSE-ONLY 3-bit 7B gate/up pair, 16.9 MB each: two per CU and op = 688 workgroups of two K slices per tile, one
      // resident round, 10.7 us against 11.0 for three per CU (1376 workgroups); at 4 bits the same cut loses, and with sparse roles in the
      // grid it is 1 % slower (profiles/r06_group_geometry.txt, r06_w3_gateup_geometry_ab.txt)
      const bool dense_only = !(op->rows && op->nnz > 0) && !(op->full_rows && op->topX > 0);
      const double two_per_cu_mb = (op->bits == 3 && dense_only) ? 18.0 : 16.0;
      target = (mb <= 12.0 ? (alone ? 2 : 1) : (!alone && mb <= two_per_cu_mb) ? 2 : (alone ? 4 : 3)) * cu_count();
      if (waves > sqllm::kWaves) target = target * sqllm::kWaves / waves;  // (16-wave workgroups: half as many, twice the rows each)
    }
    int slices = (target + gm->col_tiles / 2) / gm->col_tiles;
    if (slices < 1) slices = 1;
    if (slices > max_slices) slices = max_slices;
    upw = (gm->units_total + slices - 1) / slices;
    // ONE resident round for an op alone in its launch on a batch tile that holds two or three workgroups per CU (round 6).  The targets above
    // presume the batch-1 kernel's four: on a narrower residency the same plan is a full round plus a few stragglers, which start when the first
    // workgroups END -- 13B o_proj (4-bit, 3-6 rows, three per CU): 800 (+136 sparse) against 768 slots, 9.4 / 10.0 -> 7.6 / 9.15 us at 3 / 4 rows with
    // 560; 7B o_proj (4-bit, 7-8 rows, two per CU): 512 + 90 against 512, 11.6 / 13.3 -> 10.75 / 12.1 with 384; 13B o_proj at 3 bits (4-8 rows): 400 +
    // 136 against 512, -6...-11 % with 240 (profiles/r06_oproj_one_round.txt).  So: if dense + sparse workgroups exceed the slots by up to 35 %,
    // take the K slices that fit beside the sparse workgroups.  Not at four per CU (batch 1: 7B down_proj 960 + 242 against 1024 is 3 % FASTER than
    // 768 + 242), not for launches that are several rounds anyway, not for groups (flat: r06_launch_geometry_cols.txt, last block).
    const int bt = sqllm::batch_tile_op(gm->batch);
    const int per_cu = (bt == 1 || (bt == 2 && op->bits == 4)) ? 4 : (op->bits == 4 ? (bt <= 6 ? 3 : 2) : (bt <= 3 ? 3 : 2));
    // ... and, at four per CU, for a 3-BIT op alone in its batch-1 launch: a launch that fits gets the dense-wave priority (set_role_priority, rule 1) -- 65B
    // o_proj, 1024 + 359 against 1024: 11.8 -> 11.2 us with 512 (10.85 with 384; profiles/r06_launch_geometry_b1.txt)
    const bool w3_batch1 = per_cu == 4 && op->bits == 3 && gm->batch == 1;
    if (ops_in_launch <= 1 && (per_cu <= 3 || w3_batch1) && knobs().target_wgs.load(std::memory_order_relaxed) <= 0) {
      const int slots = per_cu * cu_count();
      const int sparse = ((op->rows && op->nnz > 0) ? (op->nnz + sqllm::kCsrChunk - 1) / sqllm::kCsrChunk : 0) +
                         ((op->full_rows && op->topX > 0) ? (op->K + sqllm::kTopxRows - 1) / sqllm::kTopxRows : 0);
      auto dense_of = [&](int sl) {
        int u = (gm->units_total + sl - 1) / sl;
        u = (u + step - 1) / step * step;
        return gm->col_tiles * ((gm->units_total + u - 1) / u);
      };
      const int total = dense_of(slices) + sparse;
      if (total > slots && 100ll * total <= (w3_batch1 ? 140ll : 135ll) * slots) {
        while (slices > 1 && dense_of(slices) + sparse > slots) --slices;
        upw = (gm->units_total + slices - 1) / slices;
      }
    }
  }
  upw = (upw + step - 1) / step * step;
  // 4-bit batch-1 waves work in chunks of four steps (128 units per workgroup, sqllm_fused.h: NBUF): a K slice of ONE chunk plus ONE step -- 160 units, the
  // 13B gate/up and down_proj under the size classes above -- pays a second round of loads for a quarter of a chunk.  One chunk per slice instead
  // (more slices): 13B s45 gate/up 22.2-22.8 -> 21.2-21.9 us, down_proj 12.85-13.5 -> 12.2-12.35 (profiles/r06_launch_geometry_b1.txt).
  if (op->bits == 4 && gm->batch == 1 && waves == sqllm::kWaves && knobs().groups_per_wave.load(std::memory_order_relaxed) <= 0 &&
      knobs().target_wgs.load(std::memory_order_relaxed) <= 0 && upw == 5 * step && (gm->units_total + 4 * step - 1) / (4 * step) <= max_slices)
    upw = 4 * step;
  gm->units_per_wg = upw;
  gm->k_slices = (gm->units_total + upw - 1) / upw;
  gm->dense_blocks = gm->col_tiles * gm->k_slices;
  gm->nnz = (op->rows && op->nnz > 0) ? op->nnz : 0;
  gm->csr_blocks = (gm->nnz + sqllm::kCsrChunk - 1) / sqllm::kCsrChunk;
  gm->topX = (op->full_rows && op->topX > 0) ? op->topX : 0;
  // top-X rows: a role of their own, one workgroup per kTopxRows k's (the fused linear folds them
  // into the dense tiles instead and drops these workgroups from its plan)
  gm->topx_blocks = gm->topX ? (op->K + sqllm::kTopxRows - 1) / sqllm::kTopxRows : 0;
  // dense blocks start at a multiple of 8 so that (dense id % 8) is the XCD of the workgroup
  gm->dense_block0 = (gm->csr_blocks + gm->topx_blocks + 7) / 8 * 8;
  gm->sparse_last = knobs().sparse_last.load(std::memory_order_relaxed);
  if (gm->sparse_last) gm->dense_block0 = gm->csr_blocks + gm->topx_blocks;  // grid = dense + sparse
  if (g_experimental.csr_ablation_bits) gm->sparse_last |= g_experimental.csr_ablation_bits() << 1;  // (measurement library)
}

// Whose waves win the issue arbitration in a BATCH-1 operator launch with sparse roles (round 6; `segs` = the launch's segments as
// make_plan left them, `total` = its workgroups).  The CSR / top-X workgroups are chains of memory round trips in front of the grid; the dense
// workgroups are issue-bound.  Same-box A/Bs of variant builds, three alternating repetitions each, on two to three boxes per rule
// (profiles/r06_dense_priority_ab.txt, r06_sparse_order_ab.txt, r06_final_ab.txt, r06_sparse_priority_ab.txt, r06_role_priority_ab.txt):
//  1 (dense waves at s_setprio 1): 3-bit launches whose workgroups are all resident at once (four 8-wave workgroups per CU) -- the sparse
//    workgroups finish before the dense tail anyway, and every issue slot they win is taken from the decode: 7B w3 s45 +2.1 / +2.5 / +3.5 %
//    tokens/s (o_proj 5.1 -> 4.7-4.9 us, down_proj 8.1 -> 7.4-8.0).  In a multi-round launch the same setting starves the sparse workgroups
//    that hold the slots the next dense workgroups wait for (gate/up +5 %), and at 4 bits it measured +-0 / -2 %.
//  2 (sparse waves at s_setprio 1: out of the way sooner): 4-bit launches (7B w4 s45 +0.9 / +2.3 %: o_proj -4...-6 %, q/k/v -2 %; 13B +0.3 %),
//    and 3-bit multi-round launches whose sparse workgroups alone are half a round or more (65B: q/k/v and gate/up -7 %, the pass +3.0 /
//    +3.3 %); NOT the smaller 3-bit multi-round launches (7B gate/up +1.3 % on both boxes).
// Built on the way and dropped: the sparse workgroups LAST in the grid per launch (-3...-9 % per launch on one box, +3...+12 % on another;
// as a global switch it costs o_proj +12 %, which is why option sparse_last measured as "no change" in rounds 3 and 5).
void set_role_priority(sqllm::Segment* segs, int n, int bits, int batch, int total, bool widened) {
  if (batch > 1 || (segs[0].gm.sparse_last & 1)) return;
  int sparse = 0;
  for (int i = 0; i < n; ++i) sparse += (segs[i].gm.csr_wide ? 2 : 1) * segs[i].gm.csr_blocks + segs[i].gm.topx_blocks;  // (counted in chunks of kCsrChunk)
  if (!sparse) return;
  const bool fits = !widened && total <= 4 * cu_count();
  int prio = 0;
  if (bits == 3) prio = fits ? 1 : (sparse >= 2 * cu_count() ? 2 : 0);
  else prio = 2;
  for (int i = 0; i < n; ++i) segs[i].gm.dense_prio = prio;
}

// CSR chunks of 2 * kCsrChunk non-zeros in operator launches of the fused kernel that exceed the resident slots AND carry many sparse workgroups (>= 1.25 x
// CUs at kCsrChunk each): half as many workgroups in front of the grid, each with twice the non-zeros in the same chain of round trips.  A build with
// 2048 everywhere (profiles/r06_sparse_granularity_ab.txt) showed both sides: gate/up -2.7...-6.4 % (7B 430 / 13B 702 sparse workgroups), 13B
// down_proj -5.7 % (365), 13B q/k/v -3 % (408), against o_proj +13...+29 % (its fewer, longer sparse workgroups become the launch's tail), 7B q/k/v
// +1-1.5 % (270) and the 7B 3-bit down_proj +13 % (fits: dense priority).  Returns true if it widened; `*total` is recomputed then.
// The batch tiles of up to 5 rows take part (three workgroups per CU there, four in the 4-bit 2-row tile): 13B s45 layer at 2 / 4 / 5 rows -1.7 / -1.1 / -1.1 % (r06_wide_chunks_tiles.txt).
bool widen_csr_chunks(sqllm::Segment* segs, int n, int bits, int batch, int* total) {
  const int resident = (batch <= 1 || (batch == 2 && bits == 4)) ? 4 : 3;  // (workgroups per CU of the tile that serves `batch` rows: sqllm_fused.h, fused_min_waves)
  if (batch > 5 || *total <= resident * cu_count() || (segs[0].gm.sparse_last & 1)) return false;
  int sparse = 0;
  for (int i = 0; i < n; ++i) sparse += segs[i].gm.csr_blocks + segs[i].gm.topx_blocks;
  if (4 * sparse < 5 * cu_count()) return false;
  int t = 0;
  for (int i = 0; i < n; ++i) {
    sqllm::KernelGeom& gm = segs[i].gm;
    if (gm.csr_blocks > 0) {
      gm.csr_wide = 1;
      gm.csr_blocks = (gm.nnz + 2 * sqllm::kCsrChunk - 1) / (2 * sqllm::kCsrChunk);
      gm.dense_block0 = (gm.csr_blocks + gm.topx_blocks + 7) / 8 * 8;
    }
    t += (gm.dense_block0 + gm.dense_blocks + 7) / 8 * 8;
  }
  *total = t;
  return true;
}

void fill_segment(const sqllm_op* op, sqllm::Segment* sg) {
  sg->q = reinterpret_cast<const uint32_t*>(op->qweight);
  sg->y = op->mul;
  sg->lut = op->lookup_table;
  sg->rows = op->rows;
  sg->cols = op->cols;
  sg->vals = op->vals;
  // (full_rows without columns is no term at all: the kernels key the top-X work on the pointer)
  sg->full_rows = op->topX > 0 ? op->full_rows : nullptr;
  sg->full_idx = op->topX > 0 ? op->full_row_indices : nullptr;
  sg->bias = nullptr;
  sg->out16 = nullptr;
}

// Geometry of the wide-batch (matrix-core) kernel: one pass covers 16 * mb batch rows (blockIdx.y
// walks the passes).  The dense work of a pass is the FLATTENED (column tile, unit) space cut into
// equal contiguous ranges, one per workgroup, as many as the chip holds at once (2 per CU; 1 for
// the 64-row kernels, by their registers) divided by the number of passes: one round of
// workgroups, none of them short (N / 64 is rarely a multiple of the CU count: cutting K slices per
// column tile left the last round 27 % full on the 13B gate/up shape).  A range is a whole number
// of workgroup steps (waves x 4 units); one that crosses a tile boundary costs a second piece.
// `reserve`: workgroups of the launch that are not dense ranges (the fused small launch's top-X slabs) -- they hold
// slots of the one round too.
void make_plan_mfma(const sqllm_op* op, sqllm::KernelGeom* gm, int ops_in_launch = 1, int row_blocks = 0, int wgs_per_cu = 0, int reserve = 0) {
  make_plan(op, gm, 1);
  const int mb = row_blocks > 0 ? row_blocks : sqllm::mfma_row_blocks(gm->batch);
  const int grid_y = (gm->batch + 16 * mb - 1) / (16 * mb);
  const int step = sqllm::kWaves * 4;
  const long long total_units = (long long)gm->col_tiles * gm->units_total;
  long long upw = (long long)knobs().groups_per_wave.load(std::memory_order_relaxed) * step;
  bool aligned = false;
  if (upw <= 0) {
    int target = knobs().target_wgs.load(std::memory_order_relaxed);
    if (target <= 0) {
      target = (wgs_per_cu > 0 ? wgs_per_cu : mb == 4 ? 1 : 2) * cu_count();
      if (reserve > 0 && target - reserve >= target / 2) target -= reserve;
    }
    target = (target + ops_in_launch - 1) / ops_in_launch;  // the ops of a group share the launch's workgroups
    long long ranges = (target + grid_y - 1) / grid_y;
    if (ranges < 1) ranges = 1;
    upw = (total_units + ranges - 1) / ranges;
    // tile-aligned ranges where a whole number per tile comes within 10 % of the wanted count (see make_plan_cols)
    const long long upws = (upw + step - 1) / step * step;
    const long long need = (gm->units_total + upws - 1) / upws;
    for (long long per_tile = need; per_tile >= 1 && per_tile >= need - 1 && !aligned; --per_tile) {
      long long even = (gm->units_total + per_tile - 1) / per_tile;
      even = (even + step - 1) / step * step;
      const long long n_even = (long long)gm->col_tiles * ((gm->units_total + even - 1) / even);
      if (n_even <= ranges && 10 * n_even >= 9 * ranges) { upw = even; aligned = true; }
    }
    // Measured exception (profiles/r06_launch_geometry_cols.txt): a 4-bit group of three ops WITH sparse terms whose aligned cut is three
    // UNEVEN ranges per tile (K = 5120: 216 / 216 / 208 units) is 5-7 % faster on two even ones -- 13B q/k/v at 3 / 4 rows 19.5 / 22.0 ->
    // 18.6 / 20.5 us (four per tile: 18.8 / 20.8; the 408 sparse workgroups hold half the slots first).  Not so dense-only, at 3 bits, or where
    // the cut is even already (7B, 65B: fewer ranges cost 5-17 % there).
    if (aligned && ops_in_launch >= 3 && op->bits == 4 && gm->nnz > 0 && (gm->units_total + upw - 1) / upw == 3 &&
        gm->units_total % upw != 0 && gm->units_total % (2 * sqllm::kWaves) == 0)
      upw = gm->units_total / 2;
  }
  upw = (upw + step - 1) / step * step;
  if (upw > 0x3fffffff) upw = 0x3fffffff / step * step;
  gm->units_per_wg = (int)upw;
  gm->k_slices = (int)((gm->units_total + upw - 1) / upw);  // pieces per column tile (reported by plan_query)
  gm->dense_blocks = aligned ? gm->col_tiles * gm->k_slices : (int)((total_units + upw - 1) / upw);
  gm->sparse_last = 0;
  gm->dense_block0 = (gm->csr_blocks + gm->topx_blocks + 7) / 8 * 8;
}

// Geometry of the WIDE matrix-core kernel (sqllm_mfma_wide.hip: sqllm_fused_wide).  A UNIT is a block of 64 rows x a
// group of 8 column tiles (one per wave).  Units are worked off one workgroup per CU at a time; as many whole rounds as
// there are run every unit over ALL of K (no atomics, one table build); the units of the last, partial round are cut
// into as many K slices as fill the idle CUs (their workgroups add atomically).  dense_blocks = workgroups of the 1-D
// grid = full_units + (units - full_units) * k_slices; units_per_wg = units of K per slice.
constexpr int kWideTiles = 8;
int make_plan_wide(const sqllm_op* op, sqllm::KernelGeom* gm) {  // returns full_units
  make_plan(op, gm, 1);
  const int row_blocks = (gm->batch + 63) / 64;
  const int col_groups = (gm->col_tiles + kWideTiles - 1) / kWideTiles;
  const long long units = (long long)col_groups * row_blocks;
  const int cus = cu_count();
  const long long full = units / cus * cus;
  const long long rem = units - full;
  int max_s = gm->units_total / (op->bits == 4 ? 32 : 8);  // a slice is at least 8 steps of 32 k's x 64 rows x 64 columns
  if (max_s > sqllm::kMaxSlices) max_s = sqllm::kMaxSlices;
  if (max_s < 1) max_s = 1;
  int s = rem > 0 ? (int)(cus / rem) : 1;
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  const int want = knobs().groups_per_wave.load(std::memory_order_relaxed) * 4;  // (option: units of K per slice, in groups of 4)
  int upw = want > 0 ? want : ((gm->units_total + s - 1) / s + 3) / 4 * 4;
  if (upw > gm->units_total) upw = (gm->units_total + 3) / 4 * 4;
  gm->units_per_wg = upw;
  gm->k_slices = (gm->units_total + upw - 1) / upw;
  const long long blocks = full + rem * gm->k_slices;
  gm->dense_blocks = blocks > 0x7fffffff ? 0x7fffffff : (int)blocks;
  gm->sparse_last = 0;
  gm->dense_block0 = (gm->csr_blocks + gm->topx_blocks + 7) / 8 * 8;
  return (int)full;
}

// When the wide form pays (13B shapes, 64-2048 rows, profiles/r04_wide_slabs.txt).  With vec as planes and the K slices'
// sums as slabs it beats the tile kernel (~175 dense-equivalent TFLOP/s at any size) once the op is ~65 us of that: 4-bit
// batch * K * N >= 5.7e9 (5120x13824: 128 rows 104 -> 85 us, 256 rows 206 -> 145, 2048 rows 1.66 -> 0.96 ms; 5120x5120: 256
// rows 75 -> 71, below that it loses), 3-bit >= 4e9 (its tile kernel is slower: 64 rows 71 -> 62).  Inside a stream capture
// the scratch costs an allocation and a free node (~25 us per group): three times that.  WITHOUT scratch (vec split in
// registers, slices add atomically) only once its units fill 80 % of the CUs (5120x13824: from 512 rows).  An explicit
// mfma_wide_min_batch is taken at its word.
bool takes_wide_path(const sqllm_op* op, bool with_scratch, bool capturing) {
  if (!knobs().mfma_split.load(std::memory_order_relaxed)) return false;
  const int from = knobs().mfma_wide_min_batch.load(std::memory_order_relaxed);
  if (from > 0) return op->batch >= from;
  if (op->batch < 64) return false;
  if (with_scratch) {
    const double work = (double)op->batch * op->K * op->N;
    return work >= (op->bits == 4 ? 5.7e9 : 4e9) * (capturing ? 3.0 : 1.0);
  }
  const long long col_groups = ((op->N + sqllm::kTileN - 1) / sqllm::kTileN + kWideTiles - 1) / kWideTiles;
  const long long units = col_groups * ((op->batch + 63) / 64);
  return 5 * units >= 4 * (long long)cu_count();
}

// The CSR term walked by the dense workgroups themselves (csr_tile_fold, sqllm_roles.h): no chunk workgroups in the grid.
void fold_csr_into_dense(sqllm::KernelGeom* gm) {
  gm->fold_csr = gm->nnz > 0 ? 1 : 0;
  gm->csr_blocks = 0;
  gm->dense_block0 = (gm->sparse_last & 1) ? gm->topx_blocks : (gm->topx_blocks + 7) / 8 * 8;
}

// top-X workgroups of an op in the fused small launch: with the transposed vec at hand 8 (K <= 40 slabs) or 16 of them
// share the op's slabs (topx_role_xt: all batch rows at once) -- few enough to hold their slots beside ONE round of
// dense workgroups; without it one workgroup per slab, as everywhere else
int small_topx_blocks(const sqllm_op* op, bool with_xT, int ops_in_launch) {
  if (!(op->full_rows && op->topX > 0)) return 0;
  const int slabs = (op->K + sqllm::kTopxRows - 1) / sqllm::kTopxRows;
  if (!with_xT || op->topX > 16) return slabs;
  // (an op alone in its launch is short-lived -- 13B o_proj: 11 us -- and leaves slots free: more, shorter-lived workgroups)
  const int want = ops_in_launch == 1 ? (slabs <= 24 ? 24 : 16) : slabs <= 40 ? 8 : 16;
  return slabs < want ? slabs : want;
}

// dense workgroups per CU the fused small launch's planner aims at: as many as the kernel holds (two, by its registers:
// capped at 80 for a third one, the dense role measured 10 % slower -- profiles/r05_small_split_register_cap.txt)
int small_wgs_per_cu_of(const sqllm_op* op) {
  const int v = knobs().small_wgs_per_cu.load(std::memory_order_relaxed);
  return v > 0 ? v : 2;
}

// rows from which an op -- or, n_ops > 1, the group it leads -- leaves the batch tiles / the column-lane kernel for the matrix cores
int mfma_min_batch_of(const sqllm_op* op, int n_ops = 1) {
  const int v = knobs().mfma_min_batch.load(std::memory_order_relaxed);
  if (v > 0) return v;  // (an explicit value is taken at its word, whatever the shape)
  // 3-bit: 17 until round 4 -- from 9 rows the fused small-batch launch of the split matrix-core kernel beats the
  // column-lane kernel (13B s45 layer 191-226 vs 270-286 us at 9-16 rows).
  if (op->bits != 4) return 9;
  // 4-bit: 5 in rounds 4-5 (the fused small launch against the 8-ROW tile, which a 5-row batch paid in full).  Round 6: batch
  // tiles of exactly 5 and 6 rows, 78 / 80 VGPRs = three workgroups per CU like the 2- / 4-row ones -- same box, 13B s45 layer
  // at 5 / 6 rows 121 / 121 us (fused small launch) -> 104 / 115 (tiles); at 7 / 8 rows the 8-row tile (96 VGPRs, two per CU)
  // loses, 150 / 155 against 122 / 123 (profiles/r06_small_batch_tiles_5_6.txt).  Except for a SMALL op alone in its launch
  // (<= 16 MB of packed weights: the 13B o_proj), whose fused small launch is mostly head, tail and the kernel in front of
  // it: 8-row tile up to 8 rows (o_proj 16.5 against 22.4 us by events).
  const double mb = (double)op->K * op->N / 2e6;
  if (n_ops == 1 && mb <= 16.0) return 9;
  return 7;
}
int cols_max_batch_of(const sqllm_op* op) {
  const int v = knobs().cols_max_batch.load(std::memory_order_relaxed);
  // 4-bit: up to 4 rows; single ops of >= 20 MB up to 6 (round 6, with the kernel's passes of exactly 5 / 6 rows: 13B down_proj 22.4 / 25.7 us against
  // 23.6 / 26.7 on the 5- / 6-row tiles, the layer -0.8 / -1.9 %; at 7 rows its 7-row pass beats the fused small launch by events, 29.6 against
  // 32.5 us, and loses by graph wall, the layer +1.1 %: profiles/r06_cols_single_ops_5_7.txt; groups are kept at 4 rows by cols_pays)
  return v > 0 ? v : (op->bits == 3 ? 16 : ((double)op->K * op->N / 2e6 >= 20.0 ? 6 : 4));
}

bool takes_mfma_path(const sqllm_op* op, int n_ops = 1) { return op->batch >= 1 && op->batch >= mfma_min_batch_of(op, n_ops); }

// does this op (or the group it leads) run as the fused small launch of the split matrix-core kernel (sqllm_fused_small_split)?
bool takes_small_split(const sqllm_op* op, int n_ops = 1) {
  if (op->K >= (1 << 26)) return false;  // (its folded CSR walk packs a local row beside the column)
  return takes_mfma_path(op, n_ops) && op->batch <= sqllm::kSmallSplitRows && knobs().mfma_split.load(std::memory_order_relaxed) &&
         knobs().mfma_fuse_small.load(std::memory_order_relaxed);
}

// Geometry of the small-batch column-lane kernel: passes of batch_tile(batch) <= 8 rows
// (blockIdx.y); the dense work of a pass is cut into equal ranges of the flattened
// (column tile, unit) space like make_plan_mfma's, three workgroups per CU (the phases of a
// workgroup -- table build, decode, combine -- hide behind its neighbours').
void make_plan_cols(const sqllm_op* op, sqllm::KernelGeom* gm, int ops_in_launch = 1) {
  make_plan(op, gm, 1);
  const int bt = sqllm::batch_tile_op(gm->batch);
  const int grid_y = (gm->batch + bt - 1) / bt;
  const long long total_units = (long long)gm->col_tiles * gm->units_total;
  long long upw = (long long)knobs().groups_per_wave.load(std::memory_order_relaxed) * sqllm::kWaves;
  bool aligned = false;
  if (upw <= 0) {
    int target = knobs().target_wgs.load(std::memory_order_relaxed);
    if (target <= 0) target = 3 * cu_count();
    target = (target + ops_in_launch - 1) / ops_in_launch;  // the ops of a group share the launch's workgroups
    long long ranges = (target + grid_y - 1) / grid_y;
    if (ranges < 1) ranges = 1;
    upw = (total_units + ranges - 1) / ranges;
    // A range that crosses a column-tile boundary is worked off as two pieces, each with its own table build.
    // Where a whole number of ranges per tile comes within 10 % of the wanted count, cut the tiles that way
    // instead (K = 5120: 640 units per tile against ranges of 200 -- two of three ranges crossed; the 5120-wide
    // ops were the one family the column-lane kernel lost on, profiles/r03_tile_vs_cols_by_shape.txt).
    const long long upw8 = (upw + sqllm::kWaves - 1) / sqllm::kWaves * sqllm::kWaves;
    const long long need = (gm->units_total + upw8 - 1) / upw8;  // ranges of that length a tile needs
    for (long long per_tile = need; per_tile >= 1 && per_tile >= need - 1 && !aligned; --per_tile) {
      long long even = (gm->units_total + per_tile - 1) / per_tile;
      even = (even + sqllm::kWaves - 1) / sqllm::kWaves * sqllm::kWaves;
      const long long n_even = (long long)gm->col_tiles * ((gm->units_total + even - 1) / even);
      if (n_even <= ranges && 10 * n_even >= 9 * ranges) { upw = even; aligned = true; }
    }
    // Measured exception (profiles/r06_launch_geometry_cols.txt): a 4-bit group of three ops WITH sparse terms whose aligned cut is three
    // UNEVEN ranges per tile (K = 5120: 216 / 216 / 208 units) is 5-7 % faster on two even ones -- 13B q/k/v at 3 / 4 rows 19.5 / 22.0 ->
    // 18.6 / 20.5 us (four per tile: 18.8 / 20.8; the 408 sparse workgroups hold half the slots first).  Not so dense-only, at 3 bits, or where
    // the cut is even already (7B, 65B: fewer ranges cost 5-17 % there).
    if (aligned && ops_in_launch >= 3 && op->bits == 4 && gm->nnz > 0 && (gm->units_total + upw - 1) / upw == 3 &&
        gm->units_total % upw != 0 && gm->units_total % (2 * sqllm::kWaves) == 0)
      upw = gm->units_total / 2;
  }
  upw = (upw + sqllm::kWaves - 1) / sqllm::kWaves * sqllm::kWaves;
  if (upw > 0x3fffffff) upw = 0x3fffffff / sqllm::kWaves * sqllm::kWaves;
  gm->units_per_wg = (int)upw;
  gm->k_slices = (int)((gm->units_total + upw - 1) / upw);
  // (the kernel recognises the tile-aligned cut by dense_blocks == col_tiles * k_slices; a contiguous cut that happens
  // to satisfy the same equation is then READ as tile-aligned -- ranges of units_per_wg units that restart at every
  // tile -- which covers every unit exactly once as well: tests/test_capi_cpu.py fuzzes both readings)
  gm->dense_blocks = aligned ? gm->col_tiles * gm->k_slices : (int)((total_units + upw - 1) / upw);
  gm->sparse_last = 0;
  gm->dense_block0 = (gm->csr_blocks + gm->topx_blocks + 7) / 8 * 8;
}

int cols_min_batch_of() {
  const int v = knobs().cols_min_batch.load(std::memory_order_relaxed);
  return v > 0 ? v : 2;
}

// Does the column-lane kernel pay?  Measured by shape and group size (profiles/r03_tile_vs_cols_by_shape.txt, hybrid
// ops, 2-16 rows), after the 2- / 4-row batch tiles went to three workgroups per CU and the column-lane kernel
// to tile-aligned ranges.  4-bit: groups of THREE ops (q/k/v: -6...-14 %) and single ops of >= 20 MB packed weights
// (down_proj: -7...-8 %); small single ops and two-op groups (gate/up, whose tile count does not cut evenly) are
// 5-19 % faster on the tiles.  3-bit: >= 16 MB packed at up to 4 rows, or many column tiles (N >= 8192); the small
// square ops stay on the tiles.  Applied only while the routing options are at their defaults: an explicit
// cols_min_batch / cols_max_batch is taken at its word.
bool cols_pays(const sqllm_op* op, int n_ops) {
  if (knobs().cols_min_batch.load(std::memory_order_relaxed) > 0 || knobs().cols_max_batch.load(std::memory_order_relaxed) > 0) return true;
  const double mb = (double)op->K * op->N * op->bits / 8e6;  // (of a group: the sum of its ops)
  // (round 6: the 4-bit 2-row tile went to four workgroups per CU -- a three-op group WITH sparse terms of >= 32 MB is then faster on the tiles at exactly
  // 2 rows: 13B q/k/v 17.1 -> 15.4 us, 65B 35 -> 32; 7B's 25 MB and every dense-only group stay here: profiles/r06_tile2_half.txt)
  if (op->bits == 4) return (n_ops >= 3 && op->batch <= 4 && !(op->batch == 2 && op->nnz > 0 && mb >= 32.0)) || (n_ops == 1 && mb >= 20.0);
  if (op->batch <= 4 && mb >= 16.0) return true;
  return op->N >= 8192;
}

// ... and at BATCH 1 (the matvec operators; a *_batched call with one row) -- round 6, second session.  The column-lane kernel's loop is 1.5 vector + 1 LDS
// instructions per weight at one row (vec in SGPRs: no broadcasts) against the fused kernel's 2.1 + 1; it had lost at batch 1 in round 3, before its ranges were cut
// tile by tile and before groups ran as one launch.  Re-measured per launch shape (tools/experiments/launch_geometry.py --batch 1 --options cols_min_batch=1
// against --batch 0; graph wall, column-lane against fused; profiles/r06_cols_batch1.txt):
//                 o_proj (alone)   down_proj (alone)   q/k/v (3 ops)   gate/up (2 ops)
//   7B  w4 s0        +12 %             -3.5 %             -11.5 %          +1.3 %
//   7B  w3 s0        +13 %             -2 %               -6.7 %           -11 %
//   13B w4 s0        +20 %             -11.5 %            -3.6 %           -2.5 %
//   13B w3 s0        +38 %             -0.2 %             -8.3 %           -8 %
//   65B w3 s0        +7.6 %            -6.7 %             -9.8 %           -1.3 %
//   7B  w4 s45       +4.5 %            +1 %               -1.7 %           +18 %
//   7B  w3 s45       +10.6 %           -2.7 %             -2.3 %           -9 %
//   13B w4 s45       +27 %             +6.5 %             +2 %             +17 %
//   13B w3 s45       +27 %             +9.7 %             +5.5 %           +6.1 %
//   65B w3 s45       +15 %             +0.7 %             -5 %             +13 %
// Dense-only launches of >= 16 MB win on it -- three-op groups always, a single op if it is tall (K >= 2 N: the square 65B o_proj loses), two-op groups at 3 bits.
// With sparse roles in the grid (which have none of the fused path's role priorities and wide chunks here) only the 3-bit groups whose K cuts into even ranges
// (K a multiple of 2048) up to 40 MB do: 7B q/k/v and gate/up -- and the 4-bit 7B q/k/v group, by a little.
bool cols_pays_batch1(const sqllm_op* sum, int n_ops, bool sparse) {
  const double mb = (double)sum->K * sum->N * sum->bits / 8e6;
  if (mb < 16.0) return false;
  if (!sparse) return n_ops >= 3 || (n_ops == 1 && sum->K >= 2ll * sum->N) || (n_ops == 2 && sum->bits == 3);
  // (4-bit: the three-op group only -- 7B q/k/v -1.7 / -3.5 % on two boxes; its gate/up loses 18 %)
  return sum->K % 2048 == 0 && mb <= 40.0 && n_ops >= (sum->bits == 3 ? 2 : 3);
}

static bool op_has_sparse(const sqllm_op* op) { return (op->rows && op->nnz > 0) || (op->full_rows && op->topX > 0); }

// the routing test shared by single ops and groups: `sum` = the op, or the group as the one op it is to the kernel (the sum of its columns)
static bool cols_route(const sqllm_op* sum, int n_ops, bool sparse) {
  const int b = sum->batch <= 0 ? 1 : sum->batch;
  const bool explicit_range = knobs().cols_min_batch.load(std::memory_order_relaxed) > 0 || knobs().cols_max_batch.load(std::memory_order_relaxed) > 0;
  if (b == 1 && !explicit_range) return cols_pays_batch1(sum, n_ops, sparse);
  return b >= cols_min_batch_of() && b <= cols_max_batch_of(sum) && cols_pays(sum, n_ops);
}

// a group of ops over one vec (q/k/v, gate/up) is judged as the one op it is to the kernel: the sum of its columns
bool group_takes_cols_path(const sqllm_op* ops, int n) {
  sqllm_op sum = ops[0];
  long long N = 0;
  bool sparse = false;
  for (int i = 0; i < n; ++i) {
    N += ops[i].N;
    sparse = sparse || op_has_sparse(&ops[i]);
  }
  sum.N = N > 0x7fffffff ? 0x7fffffff : (int)N;
  return !takes_mfma_path(&ops[0], n) && cols_route(&sum, n, sparse);
}

bool takes_cols_path(const sqllm_op* op) { return !takes_mfma_path(op) && cols_route(op, 1, op_has_sparse(op)); }

}  // namespace sqllm_host

using namespace sqllm_host;

extern "C" {

int sqllm_abi_version(void) { return SQLLM_ABI_VERSION; }

const char* sqllm_error_string(int code) {
  switch (code) {
    case SQLLM_OK: return "ok";
    case SQLLM_E_BITS: return "bits must be 3 or 4";
    case SQLLM_E_SHAPE: return "bad shape: need K % 32 == 0, N % 4 == 0, height == K/32*bits, positive dims";
    case SQLLM_E_NULL: return "a required pointer is NULL";
    case SQLLM_E_ALIGN: return "qweight / lookup_table must be 16-byte aligned";
    case SQLLM_E_SPARSE: return "inconsistent sparse operands";
    case SQLLM_E_BATCH: return "bad batch / vec_height";
    case SQLLM_E_OPTION: return "unknown option or bad value";
    case SQLLM_E_GROUP: return "ops of a group must share vec, K, bits and batch (1..4 ops per group)";
    default: break;
  }
  if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
  return "unknown sqllm error";
}

// Every option value is range-checked: switches take 0 / 1 only, counts their documented range (include/sqllm_hip.h).
// (Until round 5 any non-negative value was stored as it came, and one of them -- sparse_transpose = 2 -- switched to a
// timing-only mode that left the kernels reading an unwritten workspace; that experiment now lives behind the measurement
// library's hook, ExperimentalHooks::skip_prepare_small.)
int sqllm_set_option(const char* name, int value) {
  if (!name || value < 0) return SQLLM_E_OPTION;
  struct Opt { const char* name; std::atomic<int> Knobs::*field; int max; };
  static const Opt kOptions[] = {
      {"target_wgs", &Knobs::target_wgs, 1 << 24},
      {"groups_per_wave", &Knobs::groups_per_wave, 1 << 24},
      {"cu_count", &Knobs::cu_count, 1 << 16},  // for GPU-less planning tests
      {"sparse_last", &Knobs::sparse_last, 1},
      {"cols_groups", &Knobs::cols_groups, 1},
      {"mfma_min_batch", &Knobs::mfma_min_batch, 0x7fffffff},  // (a huge value: never)
      {"cols_min_batch", &Knobs::cols_min_batch, 0x7fffffff},
      {"cols_max_batch", &Knobs::cols_max_batch, 0x7fffffff},
      {"sparse_transpose", &Knobs::sparse_transpose, 1},
      {"scratch_in_capture", &Knobs::scratch_in_capture, 1},
      {"validate_csr", &Knobs::validate_csr, 1},
      {"mfma_split", &Knobs::mfma_split, 1},
      {"split_planes_min_batch", &Knobs::split_planes_min_batch, 0x7fffffff},
      {"mfma_wide_min_batch", &Knobs::mfma_wide_min_batch, 0x7fffffff},
      {"mfma_fuse_small", &Knobs::mfma_fuse_small, 1},
      {"mfma_fuse_sparse", &Knobs::mfma_fuse_sparse, 1},
      {"scratch_pool_threshold", &Knobs::scratch_pool_threshold, 1},
      {"small_wgs_per_cu", &Knobs::small_wgs_per_cu, 8},
      {"small_reserve_topx", &Knobs::small_reserve_topx, 1},
      {"small_planes", &Knobs::small_planes, 1},
  };
  for (const Opt& o : kOptions)
    if (!strcmp(name, o.name)) {
      if (value > o.max) return SQLLM_E_OPTION;
      (knobs().*(o.field)).store(value);
      return SQLLM_OK;
    }
  if (g_experimental.set_option) return g_experimental.set_option(name, value);  // (measurement library)
  return SQLLM_E_OPTION;
}

int sqllm_get_option(const char* name, int* value) {
  if (!name || !value) return SQLLM_E_OPTION;
  if (!strcmp(name, "target_wgs")) { *value = knobs().target_wgs.load(); return SQLLM_OK; }
  if (!strcmp(name, "groups_per_wave")) { *value = knobs().groups_per_wave.load(); return SQLLM_OK; }
  if (!strcmp(name, "cu_count")) { *value = knobs().cu_count.load(); return SQLLM_OK; }
  if (!strcmp(name, "sparse_last")) { *value = knobs().sparse_last.load(); return SQLLM_OK; }
  if (!strcmp(name, "cols_groups")) { *value = knobs().cols_groups.load(); return SQLLM_OK; }
  if (!strcmp(name, "mfma_min_batch")) { *value = knobs().mfma_min_batch.load(); return SQLLM_OK; }
  if (!strcmp(name, "cols_min_batch")) { *value = knobs().cols_min_batch.load(); return SQLLM_OK; }
  if (!strcmp(name, "cols_max_batch")) { *value = knobs().cols_max_batch.load(); return SQLLM_OK; }
  if (!strcmp(name, "sparse_transpose")) { *value = knobs().sparse_transpose.load(); return SQLLM_OK; }
  if (!strcmp(name, "scratch_in_capture")) { *value = knobs().scratch_in_capture.load(); return SQLLM_OK; }
  if (!strcmp(name, "validate_csr")) { *value = knobs().validate_csr.load(); return SQLLM_OK; }
  if (!strcmp(name, "mfma_split")) { *value = knobs().mfma_split.load(); return SQLLM_OK; }
  if (!strcmp(name, "split_planes_min_batch")) { *value = knobs().split_planes_min_batch.load(); return SQLLM_OK; }
  if (!strcmp(name, "mfma_wide_min_batch")) { *value = knobs().mfma_wide_min_batch.load(); return SQLLM_OK; }
  if (!strcmp(name, "mfma_fuse_small")) { *value = knobs().mfma_fuse_small.load(); return SQLLM_OK; }
  if (!strcmp(name, "mfma_fuse_sparse")) { *value = knobs().mfma_fuse_sparse.load(); return SQLLM_OK; }
  if (!strcmp(name, "scratch_pool_threshold")) { *value = knobs().scratch_pool_threshold.load(); return SQLLM_OK; }
  if (!strcmp(name, "small_wgs_per_cu")) { *value = knobs().small_wgs_per_cu.load(); return SQLLM_OK; }
  if (!strcmp(name, "small_reserve_topx")) { *value = knobs().small_reserve_topx.load(); return SQLLM_OK; }
  if (!strcmp(name, "small_planes")) { *value = knobs().small_planes.load(); return SQLLM_OK; }
  if (g_experimental.get_option) return g_experimental.get_option(name, value);  // (measurement library)
  return SQLLM_E_OPTION;
}

int sqllm_plan_query(const sqllm_op* op, sqllm_plan* plan) {
  if (!plan) return SQLLM_E_NULL;
  // planning needs shapes only; tolerate NULL data pointers here
  if (!op) return SQLLM_E_NULL;
  if (op->bits != 3 && op->bits != 4) return SQLLM_E_BITS;
  if (op->K <= 0 || op->N <= 0 || (op->K % 32) != 0 || (op->N % 4) != 0) return SQLLM_E_SHAPE;
  sqllm::KernelGeom gm;
  const bool mfma = takes_mfma_path(op);
  const bool wide = mfma && takes_wide_path(op, true, false);  // (as launched outside a capture, scratch at hand)
  int row_blocks = 0;
  if (wide) (void)make_plan_wide(op, &gm);
  else if (mfma) {
    make_plan_mfma(op, &gm);
    // (33-64 rows with more sparse workgroups than CUs: two 32-row passes, sparse terms in the grid -- launch_group_with_events)
    if (op->batch > 32 && op->batch <= 64 && gm.csr_blocks + gm.topx_blocks >= cu_count() && knobs().mfma_split.load(std::memory_order_relaxed) &&
        knobs().mfma_fuse_sparse.load(std::memory_order_relaxed)) {
      row_blocks = 2;
      make_plan_mfma(op, &gm, 1, 2);
    }
  }
  const bool cols = !mfma && takes_cols_path(op);
  if (cols) make_plan_cols(op, &gm);
  else if (!mfma) {
    make_plan(op, &gm);
    sqllm::Segment one;  // (as launched alone: a batch-1 launch that exceeds the resident slots may take wide CSR chunks)
    one.gm = gm;
    int total = (gm.dense_block0 + gm.dense_blocks + 7) / 8 * 8;
    (void)widen_csr_chunks(&one, 1, op->bits, op->batch, &total);
    gm = one.gm;
  }
  const bool small_split = mfma && !wide && takes_small_split(op);
  if (small_split) {  // the fused small launch: CSR term folded into the dense workgroups, top-X slabs in the grid
    // (as launched with a workspace: vec transposed, 8-16 top-X workgroups, the dense ranges planned beside them)
    const bool with_xT = knobs().sparse_transpose.load(std::memory_order_relaxed) != 0;
    make_plan_mfma(op, &gm, 1, 0, small_wgs_per_cu_of(op),
                   (with_xT || knobs().small_reserve_topx.load(std::memory_order_relaxed)) ? (small_topx_blocks(op, with_xT, 1) + 7) / 8 * 8 : 0);
    gm.topx_blocks = small_topx_blocks(op, with_xT, 1);
    fold_csr_into_dense(&gm);
  }
  plan->col_tiles = gm.col_tiles;
  plan->k_slices = gm.k_slices;
  plan->groups_per_wave = gm.units_per_wg;
  plan->dense_blocks = gm.dense_blocks;
  plan->csr_blocks = gm.csr_blocks;
  plan->topx_blocks = gm.topx_blocks;
  plan->grid_x = (mfma && !small_split) ? gm.dense_blocks : gm.dense_block0 + gm.dense_blocks;  // (wide batches: the sparse terms are a launch of their own)
  const int rows_per_pass = wide ? (gm.batch > 0 ? gm.batch : 1) : mfma ? 16 * (row_blocks ? row_blocks : sqllm::mfma_row_blocks(gm.batch)) : sqllm::batch_tile_op(gm.batch);
  plan->grid_y = (gm.batch + rows_per_pass - 1) / rows_per_pass;
  return SQLLM_OK;
}

static int64_t align16(int64_t v) { return (v + 15) / 16 * 16; }

// workspace of a fused linear: 64-bit accumulator words [batch, N]
int64_t sqllm_linear_workspace_bytes(const sqllm_op* op) {
  if (!op || op->N <= 0) return 0;
  return align16(8ll * (op->batch <= 0 ? 1 : op->batch) * op->N);
}

// Stream-ordered scratch of a wide-batch group (hipMallocAsync / hipFreeAsync on the caller's stream: no host
// synchronisation, the pool keeps the block for the next call; inside a stream capture the allocation and the free become
// memory nodes of the graph -- works under torch's graph capture on ROCm 7.2; option scratch_in_capture = 0 keeps
// captures allocation-free).  ONE block per group holds
//   xT     vec TRANSPOSED (xT[k][row]) for the CSR role: one coalesced read per non-zero serves every batch row instead
//          of `batch` gathers 4 K bytes apart (only with a CSR term; option sparse_transpose);
//   planes vec split once into bf16 planes in fragment order + the lo flags (sqllm_mfma_wide.hip: sqllm_split_vec) for
//          the wide form (from split_planes_min_batch rows, default 64);
//   slabs  the sums of the wide form's K slices (sqllm_wide_reduce adds them to mul).
// Without scratch: the CSR role gathers from vec, the wide form splits in registers and its slices add atomically.
constexpr int kSplitPlanesMinBatch = 64;
struct WideScratch {
  void* block = nullptr;  // stream-ordered allocation of our own (freed by the destructor), null when the caller's workspace serves
  float* xT = nullptr;
  int Bp = 0;
  void* planes = nullptr;
  uint32_t* flags = nullptr;
  float* slabs = nullptr;
  bool capturing = false;  // ... and the scratch would have to be allocated inside the capture (memory nodes)
  hipStream_t s = nullptr;
  // what a group wants, in bytes: [xT | planes | flags | slabs]
  struct Layout { uint64_t xt = 0, planes = 0, flags = 0, slabs = 0; int Bp = 0; uint64_t total() const { return xt + planes + flags + slabs; } };
  static Layout layout(const sqllm_op* ops, int n, bool capturing) {
    Layout L;
    if (!ops || n < 1 || ops[0].batch <= 0 || ops[0].K <= 0) return L;
    bool any_csr = false, any_wide = false;
    uint64_t slab_bytes = 0;
    for (int i = 0; i < n; ++i) {
      any_csr = any_csr || (ops[i].nnz > 0 && ops[i].rows && ops[i].cols && ops[i].vals);
      if (validate(&ops[i]) != SQLLM_OK || !takes_wide_path(&ops[i], true, capturing)) continue;  // (the launch loop reports errors)
      any_wide = true;
      sqllm::KernelGeom gm;  // room for the K slices' sums (the ops of a group run one after the other: the largest need serves all)
      const int full = make_plan_wide(&ops[i], &gm);
      const uint64_t b = sqllm::wide_slab_bytes(gm.dense_blocks, full, gm.k_slices);
      if (b > slab_bytes) slab_bytes = b;
    }
    const bool want_xT = any_csr && knobs().sparse_transpose.load(std::memory_order_relaxed);
    int from = knobs().split_planes_min_batch.load(std::memory_order_relaxed);
    if (from == 0) from = kSplitPlanesMinBatch;
    const uint64_t chunks = sqllm::split_planes_chunks(ops[0].batch, ops[0].K);
    const bool want_planes = any_wide && ops[0].batch >= from && chunks < (1ull << 31);  // (32-bit chunk numbers in the kernel)
    if (!want_xT && !want_planes) return L;
    L.Bp = (ops[0].batch + 63) / 64 * 64;
    L.xt = want_xT ? (uint64_t)ops[0].K * L.Bp * sizeof(float) : 0;
    L.planes = want_planes ? chunks * 16 : 0;
    L.flags = want_planes ? (sqllm::kSplitFlagWgs * sizeof(uint32_t) + 15) / 16 * 16 : 0;
    L.slabs = want_planes ? slab_bytes : 0;
    return L;
  }
  // ops[0] is validated; *e0 (the start event of a profiled group) goes to the first kernel enqueued here, if any.
  // ws / ws_bytes: the caller's workspace (sqllm_launch_*_ws), used instead of an allocation when it is large enough.
  int acquire(const sqllm_op* ops, int n, sqllm_stream_t stream, hipEvent_t* e0, void* ws, int64_t ws_bytes) {
    s = static_cast<hipStream_t>(stream);
    if (!ops[0].vec || ops[0].batch <= 0 || ops[0].K <= 0) return SQLLM_OK;
    bool in_capture = false;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (!(hipStreamIsCapturing(s, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone)) {
      (void)hipGetLastError();
      in_capture = true;
    }
    char* p = nullptr;
    Layout L = layout(ops, n, false);
    if (ws && (reinterpret_cast<uintptr_t>(ws) & 15u) == 0 && L.total() > 0 && (uint64_t)(ws_bytes > 0 ? ws_bytes : 0) >= L.total()) {
      p = static_cast<char*>(ws);  // nothing allocated, nothing to free: a capture stays free of memory nodes
    } else {
      capturing = in_capture;
      if (capturing && !knobs().scratch_in_capture.load(std::memory_order_relaxed)) return SQLLM_OK;
      L = layout(ops, n, capturing);
      if (L.total() == 0) return SQLLM_OK;
      keep_scratch_in_pool();
      if (hipMallocAsync(&block, L.total(), s) != hipSuccess || !block) {
        (void)hipGetLastError();  // no scratch
        block = nullptr;
        return SQLLM_OK;
      }
      p = static_cast<char*>(block);
    }
    if (L.total() == 0) return SQLLM_OK;
    if (L.xt) {
      xT = reinterpret_cast<float*>(p);
      Bp = L.Bp;
      const hipError_t e = sqllm::transpose_vec(ops[0].vec, xT, ops[0].batch, ops[0].K, Bp, s, e0 ? *e0 : nullptr);
      if (e != hipSuccess) return static_cast<int>(e);  // (the destructor frees)
      if (e0) *e0 = nullptr;
    }
    if (L.planes) {
      planes = p + L.xt;
      flags = reinterpret_cast<uint32_t*>(p + L.xt + L.planes);
      if (L.slabs) slabs = reinterpret_cast<float*>(p + L.xt + L.planes + L.flags);
      const hipError_t e = sqllm::split_vec(ops[0].vec, planes, flags, ops[0].batch, ops[0].K, s, e0 ? *e0 : nullptr);
      if (e != hipSuccess) return static_cast<int>(e);
      if (e0) *e0 = nullptr;
    }
    return SQLLM_OK;
  }
  ~WideScratch() {
    if (block) (void)hipFreeAsync(block, s);
  }
};

// One kernel over 1..kMaxSegments ops that share vec, K, bits and batch.  `lin` (optional) points at
// the fused-linear descriptors the ops were taken from: `ops` is then lin[i].op.
// `ws` / `ws_bytes`: the caller's workspace (sqllm_launch_*_ws; sqllm_workspace_bytes says how much a group can use), or null.
// `ws_entry`: the call came through a `_ws` entry point -- up to 16 rows nothing is allocated then, workspace or not.
static int launch_group_with_events(const sqllm_op* ops, int n, sqllm_stream_t stream, hipEvent_t e0,
                                    hipEvent_t e1, const sqllm_linear* lin = nullptr, void* ws = nullptr, int64_t ws_bytes = 0,
                                    bool ws_entry = false) {
  if (n < 1 || n > sqllm::kMaxSegments) return SQLLM_E_GROUP;
  if (!ops && !lin) return SQLLM_E_NULL;
  // (small batches: a group whose summed columns pass the column-lane kernel's test takes that kernel as ONE launch --
  // make_plan_cols divides the workgroup target by the number of ops; other groups stay on the batch tiles, which beat
  // one column-lane launch per op -- 13B s45 decoder layer at 2 rows: 88 vs 101 us)
  if (!lin && n > 1 && knobs().cols_groups.load(std::memory_order_relaxed) && group_takes_cols_path(ops, n)) {
    // a group on the column-lane kernel: ONE launch, the workgroups divided between the ops
    sqllm::LaunchArgs a;
    a.ev_start = e0;
    a.ev_stop = e1;
    a.x = ops[0].vec;
    a.ga.n_seg = n;
    memset(a.ga.seg, 0, sizeof(a.ga.seg));
    int block = 0;
    for (int i = 0; i < n; ++i) {
      const sqllm_op* op = &ops[i];
      int rc = validate(op);
      if (rc == SQLLM_OK) rc = validate_csr_values(op, stream);
      if (rc != SQLLM_OK) return rc;
      if (op->vec != ops[0].vec || op->K != ops[0].K || op->bits != ops[0].bits || op->batch != ops[0].batch)
        return SQLLM_E_GROUP;
      if ((uint64_t)op->batch * (uint64_t)op->K >= (1ull << 31)) return SQLLM_E_SHAPE;  // 32-bit row offsets into vec
      sqllm::Segment& sg = a.ga.seg[i];
      sg.q = reinterpret_cast<const uint32_t*>(op->qweight);
      sg.y = op->mul;
      sg.lut = op->lookup_table;
      sg.rows = op->rows;
      sg.cols = op->cols;
      sg.vals = op->vals;
      sg.full_rows = op->topX > 0 ? op->full_rows : nullptr;
      sg.full_idx = op->topX > 0 ? op->full_row_indices : nullptr;
      make_plan_cols(op, &sg.gm, n);
      a.ga.block0[i] = block;
      block += (sg.gm.dense_block0 + sg.gm.dense_blocks + 7) / 8 * 8;
    }
    for (int i = n; i <= sqllm::kMaxSegments; ++i) a.ga.block0[i] = block;
    return static_cast<int>(sqllm::launch_batched_cols(ops[0].bits, a, static_cast<hipStream_t>(stream)));
  }
  if (!lin && takes_small_split(&ops[0], n)) {
    // up to 16 rows on the split matrix-core kernel: ONE launch for the whole group, sparse roles included
    sqllm::LaunchArgs a;
    a.ev_start = e0;
    a.ev_stop = e1;
    a.x = ops[0].vec;
    a.ga.n_seg = n;
    memset(a.ga.seg, 0, sizeof(a.ga.seg));
    int block = 0;
    // The dense ranges are ONE round of workgroups, as many as the chip holds at once: the top-X slabs (8 per op, padded)
    // take their slots from the same count.  (Planned beside them, 20-50 dense workgroups of a 13B launch found no slot,
    // started when the first ones left and ran as a second round of their own: 34 instead of 22 us per down_proj launch at
    // 16 rows -- profiles/r05_small_split_timeline.txt.)
    // With a workspace: vec transposed first (xT[k][rows]: the folded CSR walk and the top-X slabs then read ONE line per
    // k for all the batch rows); without, they gather from vec itself.
    bool any_sparse = false;
    for (int i = 0; i < n; ++i) any_sparse = any_sparse || (ops[i].nnz > 0 && ops[i].rows) || (ops[i].full_rows && ops[i].topX > 0);
    const bool want_xT = any_sparse && ops[0].batch > 0 && ops[0].K > 0 && ops[0].vec && (reinterpret_cast<uintptr_t>(ops[0].vec) & 15u) == 0 &&
                         knobs().sparse_transpose.load(std::memory_order_relaxed);  // (the transposition reads vec 16 bytes at a time)
    // (vec transposed for the sparse terms, then its bf16 planes for the dense term: sqllm_prepare_small writes both)
    const int64_t xt_only = want_xT ? sqllm::transpose_small_bytes(ops[0].batch, ops[0].K) : 0;
    const bool want_planes = want_xT && knobs().small_planes.load(std::memory_order_relaxed);
    const int64_t xt_bytes = xt_only + (want_planes ? sqllm::small_planes_bytes(ops[0].K) : 0);
    float* xT = nullptr;
    struct Scratch {  // the workspace-less names only: stream-ordered scratch, as for the wider batches (never inside a capture:
      void* p = nullptr;  // its memory nodes cost more than the gathers -- profiles/r04_small_batch_layer.txt; never for a `_ws`
                          // entry point: those allocate nothing up to 16 rows and leave the default pool alone -- the walk gathers)
      hipStream_t s = nullptr;
      ~Scratch() { if (p) (void)hipFreeAsync(p, s); }
    } own;
    if (want_xT && ws && (reinterpret_cast<uintptr_t>(ws) & 15u) == 0 && ws_bytes >= xt_bytes) {
      xT = static_cast<float*>(ws);
    } else if (want_xT && !ws_entry) {
      own.s = static_cast<hipStream_t>(stream);
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(own.s, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone) {
        keep_scratch_in_pool();
        if (hipMallocAsync(&own.p, (size_t)xt_bytes, own.s) == hipSuccess && own.p) xT = static_cast<float*>(own.p);
        else { (void)hipGetLastError(); own.p = nullptr; }
      } else {
        (void)hipGetLastError();
      }
    }
    const bool with_xT = xT != nullptr;
    int reserve = 0;
    if (with_xT || knobs().small_reserve_topx.load(std::memory_order_relaxed))
      for (int i = 0; i < n; ++i) reserve += (small_topx_blocks(&ops[i], with_xT, n) + 7) / 8 * 8;
    for (int i = 0; i < n; ++i) {
      const sqllm_op* op = &ops[i];
      int rc = validate(op);
      if (rc == SQLLM_OK) rc = validate_csr_values(op, stream);
      if (rc != SQLLM_OK) return rc;
      if (op->vec != ops[0].vec || op->K != ops[0].K || op->bits != ops[0].bits || op->batch != ops[0].batch) return SQLLM_E_GROUP;
      if ((uint64_t)op->batch * (uint64_t)op->K >= (1ull << 31)) return SQLLM_E_SHAPE;  // 32-bit row offsets into vec
      sqllm::Segment& sg = a.ga.seg[i];
      fill_segment(op, &sg);
      make_plan_mfma(op, &sg.gm, n, 0, small_wgs_per_cu_of(op), reserve);
      sg.gm.topx_blocks = small_topx_blocks(op, with_xT, n);
      fold_csr_into_dense(&sg.gm);
      a.ga.block0[i] = block;
      block += (sg.gm.dense_block0 + sg.gm.dense_blocks + 7) / 8 * 8;
    }
    for (int i = n; i <= sqllm::kMaxSegments; ++i) a.ga.block0[i] = block;
    if (with_xT) {
      if (!(g_experimental.skip_prepare_small && g_experimental.skip_prepare_small())) {  // (measurement library only: what the kernel in front costs)
        const hipError_t e = want_planes
                                 ? sqllm::prepare_small(ops[0].vec, xT, reinterpret_cast<char*>(xT) + xt_only, ops[0].batch, ops[0].K, static_cast<hipStream_t>(stream), e0)
                                 : sqllm::transpose_small(ops[0].vec, xT, ops[0].batch, ops[0].K, static_cast<hipStream_t>(stream), e0);
        if (e != hipSuccess) return static_cast<int>(e);
        a.ev_start = nullptr;
      }
      a.xT = xT;
      if (want_planes) a.planes = reinterpret_cast<char*>(xT) + xt_only;
    }
    if (g_experimental.decorate) g_experimental.decorate(&a);  // (measurement library: timeline buffer)
    return static_cast<int>(sqllm::launch_small_split(ops[0].bits, a, static_cast<hipStream_t>(stream)));
  }
  if (!lin && (takes_mfma_path(&ops[0], n) || (n == 1 && takes_cols_path(&ops[0])))) {
    // batched operators: one launch per op (the members of a group only share their input) of the
    // matrix-core kernel (wide batches) or of the column-lane kernel (small ones)
    const bool mfma = takes_mfma_path(&ops[0], n);
    WideScratch wsc;  // (transposed vec for the CSR role, planes + slabs for the wide form: see the struct)
    if (mfma) {
      int rc = validate(&ops[0]);  // (its kernels read vec by batch and K: shape errors first)
      if (rc != SQLLM_OK) return rc;
      rc = wsc.acquire(ops, n, stream, &e0, ws, ws_bytes);
      if (rc != SQLLM_OK) return rc;
    }
    float* const xT = wsc.xT;
    const int Bp = wsc.Bp;
    for (int i = 0; i < n; ++i) {
      const sqllm_op* op = &ops[i];
      int rc = validate(op);
      if (rc == SQLLM_OK) rc = validate_csr_values(op, stream);
      if (rc != SQLLM_OK) return rc;
      if (op->vec != ops[0].vec || op->K != ops[0].K || op->bits != ops[0].bits || op->batch != ops[0].batch)
        return SQLLM_E_GROUP;
      if ((uint64_t)op->batch * (uint64_t)op->K >= (1ull << 31)) return SQLLM_E_SHAPE;  // 32-bit row offsets into vec
      sqllm::LaunchArgs a;
      a.ev_start = i == 0 ? e0 : nullptr;
      a.ev_stop = i == n - 1 ? e1 : nullptr;
      a.x = op->vec;
      a.xT = (op->nnz > 0 && op->rows && op->cols && op->vals) ? xT : nullptr;
      a.Bp = Bp;
      a.planes = wsc.planes;
      a.plane_flags = wsc.flags;
      a.wide_slabs = wsc.slabs;
      a.ga.n_seg = 1;
      memset(a.ga.seg, 0, sizeof(a.ga.seg));
      sqllm::Segment& sg = a.ga.seg[0];
      sg.q = reinterpret_cast<const uint32_t*>(op->qweight);
      sg.y = op->mul;
      sg.lut = op->lookup_table;
      sg.rows = op->rows;
      sg.cols = op->cols;
      sg.vals = op->vals;
      sg.full_rows = op->topX > 0 ? op->full_rows : nullptr;
      sg.full_idx = op->topX > 0 ? op->full_row_indices : nullptr;
      a.wide = mfma && takes_wide_path(op, wsc.planes != nullptr, wsc.capturing);
      if (a.wide) a.wide_full_units = make_plan_wide(op, &sg.gm);
      else if (mfma) make_plan_mfma(op, &sg.gm);
      else make_plan_cols(op, &sg.gm);
      a.ga.block0[0] = 0;
      for (int j = 1; j <= sqllm::kMaxSegments; ++j) a.ga.block0[j] = sg.gm.dense_block0 + sg.gm.dense_blocks;
      if (mfma) {
        // the sparse terms first, as a launch of their own (see sqllm_sparse_batched), then the dense term
        // (running the two on different streams was tried: they do not overlap -- the dense kernel holds
        // every CU's registers -- and the two event waits cost 14 us per op)
        const bool sparse = sg.gm.csr_blocks + sg.gm.topx_blocks > 0;
        // tile form: the sparse terms ride in the dense launch's grid (sqllm_fused_batched_split_all) -- always up to 32 rows
        // (two workgroups per CU: 13B shapes 59-69 -> 55-57 us, 5120x5120 35-46 -> 28-38); from 33 rows, where the kernel
        // takes a whole CU per workgroup, only while the sparse workgroups are fewer than the CUs (5120x5120 at 64 rows
        // 81 -> 54 us; with 331 of them, 5120x13824, 112 -> 117: profiles/r04_mid_rows_fused_sparse.txt)
        const bool may_fuse = sparse && !a.wide && knobs().mfma_split.load(std::memory_order_relaxed) &&
                              knobs().mfma_fuse_sparse.load(std::memory_order_relaxed);
        bool fuse_sparse = may_fuse && (op->batch <= 32 || sg.gm.csr_blocks + sg.gm.topx_blocks < cu_count());
        if (may_fuse && !fuse_sparse && op->batch <= 64) {
          // 33-64 rows with more sparse workgroups than CUs: two passes of 32 rows on the kernel that leaves room for two
          // workgroups per CU, sparse terms in its grid, instead of one 64-row pass + their own launch
          // (profiles/r04_mid_rows_fused_sparse.txt, "mb2")
          a.row_blocks = 2;
          make_plan_mfma(op, &sg.gm, 1, 2);
          for (int j = 1; j <= sqllm::kMaxSegments; ++j) a.ga.block0[j] = sg.gm.dense_block0 + sg.gm.dense_blocks;
          fuse_sparse = true;
        }
        if (fuse_sparse) {
          rc = static_cast<int>(sqllm::launch_batched_mfma_split_all(op->bits, a, static_cast<hipStream_t>(stream)));
          if (rc != SQLLM_OK) return rc;
          continue;
        }
        if (sparse) {
          sqllm::LaunchArgs as = a;
          as.ev_stop = nullptr;
          if (g_experimental.decorate) g_experimental.decorate(&as);  // (measurement library: timeline probe, tools/timeline.py --batch)
          rc = static_cast<int>(sqllm::launch_batched_sparse(as, static_cast<hipStream_t>(stream)));
          if (rc != SQLLM_OK) return rc;
          a.ev_start = nullptr;
        }
        rc = static_cast<int>(knobs().mfma_split.load(std::memory_order_relaxed)
                                  ? sqllm::launch_batched_mfma_split(op->bits, a, static_cast<hipStream_t>(stream))
                                  : sqllm::launch_batched_mfma(op->bits, a, static_cast<hipStream_t>(stream)));
      } else {
        rc = static_cast<int>(sqllm::launch_batched_cols(op->bits, a, static_cast<hipStream_t>(stream)));
      }
      if (rc != SQLLM_OK) return rc;
    }
    return SQLLM_OK;
  }
  if (!lin && g_experimental.route) {  // (measurement library: the measured-and-not-adopted batch-1 kernels)
    int rc = SQLLM_OK;
    if (g_experimental.route(ops, n, stream, e0, e1, &rc)) return rc;
  }
  sqllm_op tmp[sqllm::kMaxSegments];
  if (lin) {
    for (int i = 0; i < n; ++i) {
      tmp[i] = lin[i].op;
      if (!lin[i].workspace) return SQLLM_E_NULL;
      if ((reinterpret_cast<uintptr_t>(lin[i].workspace) & 15u) != 0) return SQLLM_E_ALIGN;
    }
    ops = tmp;
  }
  sqllm::LaunchArgs a;
  a.linear = lin != nullptr;
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
    if (op->vec != ops[0].vec || op->K != ops[0].K || op->bits != ops[0].bits ||
        (op->batch <= 0 ? 1 : op->batch) != (ops[0].batch <= 0 ? 1 : ops[0].batch))
      return SQLLM_E_GROUP;
    sqllm::Segment& sg = a.ga.seg[i];
    fill_segment(op, &sg);
    // fused linear: a column's K slices + the CSR chunks its row can be spread over must fit the 6-bit count
    const int csr_bound = (op->rows && op->nnz > 0) ? op->K / sqllm::kCsrChunk + 2 : 0;
    make_plan(op, &sg.gm, n, lin ? (sqllm::kMaxContrib - csr_bound > 1 ? sqllm::kMaxContrib - csr_bound : 1) : sqllm::kMaxSlices);
    if (lin) {
      // accumulate into the workspace plane; op->mul is the fp16 result
      sg.y = reinterpret_cast<float*>(lin[i].workspace);
      sg.out16 = op->mul;
      sg.bias = lin[i].bias;
      // the top-X rows are always handled inside the dense workgroups: no top-X role in the grid
      sg.gm.topx_blocks = 0;
      if (!(sg.gm.sparse_last & 1)) sg.gm.dense_block0 = (sg.gm.csr_blocks + 7) / 8 * 8;
      else sg.gm.dense_block0 = sg.gm.csr_blocks;
      // the 55-bit sum field holds at most kMaxContrib clamped contributions per column: the K
      // slices and one per CSR chunk a row can be spread over
      if (sg.gm.k_slices + (sg.gm.csr_blocks ? op->K / sqllm::kCsrChunk + 2 : 0) > sqllm::kMaxContrib)
        return SQLLM_E_SHAPE;
    }
    a.ga.block0[i] = block;
    // pad every segment to a multiple of 8 workgroups: dense ids keep their XCD alignment
    block += (sg.gm.dense_block0 + sg.gm.dense_blocks + 7) / 8 * 8;
  }
  for (int i = n; i <= sqllm::kMaxSegments; ++i) a.ga.block0[i] = block;
  for (int i = n; i < sqllm::kMaxSegments; ++i) memset(&a.ga.seg[i], 0, sizeof(sqllm::Segment));
  if (!lin) {
    const bool widened = widen_csr_chunks(a.ga.seg, n, ops[0].bits, ops[0].batch, &block);
    if (widened) {
      int at = 0;
      for (int i = 0; i < n; ++i) {
        a.ga.block0[i] = at;
        at += (a.ga.seg[i].gm.dense_block0 + a.ga.seg[i].gm.dense_blocks + 7) / 8 * 8;
      }
      for (int i = n; i <= sqllm::kMaxSegments; ++i) a.ga.block0[i] = at;
    }
    set_role_priority(a.ga.seg, n, ops[0].bits, ops[0].batch, block, widened);
  }
  if (g_experimental.decorate) g_experimental.decorate(&a);  // (measurement library: ablation bits, LDS pad, timeline buffer)
  return static_cast<int>(sqllm::launch_fused(ops[0].bits, a, static_cast<hipStream_t>(stream)));
}

int sqllm_linear_f16(const sqllm_linear* lin, sqllm_stream_t stream) {
  if (!lin) return SQLLM_E_NULL;
  return launch_group_with_events(nullptr, 1, stream, nullptr, nullptr, lin);
}

int sqllm_linear_f16_groups(const sqllm_linear* lins, const int32_t* group_sizes, int32_t n_groups,
                            sqllm_stream_t stream, int32_t* n_done) {
  if (n_done) *n_done = 0;
  if (n_groups < 0 || (n_groups > 0 && (!lins || !group_sizes))) return SQLLM_E_NULL;
  int32_t at = 0;
  for (int32_t g = 0; g < n_groups; ++g) {
    if (group_sizes[g] < 1) return SQLLM_E_GROUP;
    int rc = launch_group_with_events(nullptr, group_sizes[g], stream, nullptr, nullptr, lins + at);
    if (rc != SQLLM_OK) return rc;
    at += group_sizes[g];
    if (n_done) *n_done = g + 1;
  }
  return SQLLM_OK;
}

int sqllm_launch(const sqllm_op* op, sqllm_stream_t stream) {
  return launch_group_with_events(op, 1, stream, nullptr, nullptr);
}

int sqllm_launch_group(const sqllm_op* ops, int32_t n_ops, sqllm_stream_t stream) {
  return launch_group_with_events(ops, n_ops, stream, nullptr, nullptr);
}

int sqllm_launch_groups(const sqllm_op* ops, const int32_t* group_sizes, int32_t n_groups,
                        sqllm_stream_t stream, int32_t* n_done) {
  if (n_done) *n_done = 0;
  if (n_groups < 0 || (n_groups > 0 && (!ops || !group_sizes))) return SQLLM_E_NULL;
  int32_t at = 0;
  for (int32_t g = 0; g < n_groups; ++g) {
    int rc = launch_group_with_events(ops + at, group_sizes[g], stream, nullptr, nullptr);
    if (rc != SQLLM_OK) return rc;
    at += group_sizes[g];
    if (n_done) *n_done = g + 1;
  }
  return SQLLM_OK;
}

int64_t sqllm_workspace_bytes(const sqllm_op* ops, int32_t n_ops) {
  if (!ops || n_ops < 1 || ops[0].batch < 2 || ops[0].K <= 0) return 0;  // (batch 1: one kernel, no scratch)
  int64_t need = 0;
  if (takes_small_split(&ops[0], n_ops)) {
    bool any_sparse = false;
    for (int i = 0; i < n_ops; ++i) any_sparse = any_sparse || (ops[i].nnz > 0 && ops[i].rows) || (ops[i].full_rows && ops[i].topX > 0);
    if (any_sparse && knobs().sparse_transpose.load(std::memory_order_relaxed))
      need = sqllm::transpose_small_bytes(ops[0].batch, ops[0].K) + (knobs().small_planes.load(std::memory_order_relaxed) ? sqllm::small_planes_bytes(ops[0].K) : 0);
  } else if (takes_mfma_path(&ops[0], n_ops)) {
    need = (int64_t)WideScratch::layout(ops, n_ops, false).total();
  }
  return (need + 255) / 256 * 256;
}

int sqllm_launch_ws(const sqllm_op* op, void* workspace, int64_t workspace_bytes, sqllm_stream_t stream) {
  return launch_group_with_events(op, 1, stream, nullptr, nullptr, nullptr, workspace, workspace_bytes, true);
}

int sqllm_launch_group_ws(const sqllm_op* ops, int32_t n_ops, void* workspace, int64_t workspace_bytes, sqllm_stream_t stream) {
  return launch_group_with_events(ops, n_ops, stream, nullptr, nullptr, nullptr, workspace, workspace_bytes, true);
}

int sqllm_launch_groups_ws(const sqllm_op* ops, const int32_t* group_sizes, int32_t n_groups, void* workspace,
                           int64_t workspace_bytes, sqllm_stream_t stream, int32_t* n_done) {
  if (n_done) *n_done = 0;
  if (n_groups < 0 || (n_groups > 0 && (!ops || !group_sizes))) return SQLLM_E_NULL;
  int32_t at = 0;
  for (int32_t g = 0; g < n_groups; ++g) {
    int rc = launch_group_with_events(ops + at, group_sizes[g], stream, nullptr, nullptr, nullptr, workspace, workspace_bytes, true);
    if (rc != SQLLM_OK) return rc;
    at += group_sizes[g];
    if (n_done) *n_done = g + 1;
  }
  return SQLLM_OK;
}

int sqllm_launch_sequence(const sqllm_op* ops, int32_t n_ops, sqllm_stream_t stream, int32_t* n_done) {
  if (n_done) *n_done = 0;
  if (n_ops < 0 || (n_ops > 0 && !ops)) return SQLLM_E_NULL;
  for (int32_t i = 0; i < n_ops; ++i) {
    int rc = sqllm_launch(&ops[i], stream);
    if (rc != SQLLM_OK) return rc;
    if (n_done) *n_done = i + 1;
  }
  return SQLLM_OK;
}

int sqllm_profile_groups(const sqllm_op* ops, const int32_t* group_sizes, int32_t n_groups,
                         sqllm_stream_t stream, int32_t reps, float* avg_us) {
  return sqllm_profile_groups_ws(ops, group_sizes, n_groups, nullptr, 0, stream, reps, avg_us);
}

int sqllm_profile_groups_ws(const sqllm_op* ops, const int32_t* group_sizes, int32_t n_groups, void* workspace,
                            int64_t workspace_bytes, sqllm_stream_t stream, int32_t reps, float* avg_us) {
  if (n_groups < 0 || reps < 1 || (n_groups > 0 && (!ops || !group_sizes || !avg_us))) return SQLLM_E_NULL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipEvent_t* ev = new hipEvent_t[2 * (size_t)n_groups + 1];
  int rc = SQLLM_OK;
  int made = 0;
  for (; made < 2 * n_groups; ++made)
    if (hipEventCreate(&ev[made]) != hipSuccess) { rc = (int)hipGetLastError(); break; }
  for (int i = 0; i < n_groups; ++i) avg_us[i] = 0.f;
  for (int r = 0; r < reps && rc == SQLLM_OK; ++r) {
    int32_t at = 0;
    for (int g = 0; g < n_groups && rc == SQLLM_OK; ++g) {
      rc = launch_group_with_events(ops + at, group_sizes[g], stream, ev[2 * g], ev[2 * g + 1], nullptr, workspace, workspace_bytes);
      at += group_sizes[g];
    }
    if (rc != SQLLM_OK) break;
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) { rc = (int)e; break; }
    for (int g = 0; g < n_groups; ++g) {
      float ms = 0.f;
      e = hipEventElapsedTime(&ms, ev[2 * g], ev[2 * g + 1]);
      if (e != hipSuccess) { rc = (int)e; break; }
      avg_us[g] += ms * 1000.f;
    }
  }
  for (int g = 0; g < n_groups; ++g) avg_us[g] /= (float)reps;
  for (int i = 0; i < made; ++i) (void)hipEventDestroy(ev[i]);
  delete[] ev;
  return rc;
}

int sqllm_profile_sequence(const sqllm_op* ops, int32_t n_ops, sqllm_stream_t stream, int32_t reps,
                           float* avg_us) {
  if (n_ops < 0) return SQLLM_E_NULL;
  int32_t* ones = new int32_t[n_ops > 0 ? n_ops : 1];
  for (int32_t i = 0; i < n_ops; ++i) ones[i] = 1;
  int rc = sqllm_profile_groups(ops, ones, n_ops, stream, reps, avg_us);
  delete[] ones;
  return rc;
}

// ---- the reference operator names -------------------------------------------------------------

static int named(int bits, int batch, int vec_height, const float* vec, const int32_t* mat,
                 float* mul, const float* lut, int height, int width, const int32_t* rows,
                 const int32_t* cols, const float* vals, int nnz, int num_rows,
                 const float* full_rows, const int32_t* full_idx, int topX, bool has_csr,
                 bool has_topx, sqllm_stream_t stream) {
  if (bits != 3 && bits != 4) return SQLLM_E_BITS;
  if (height <= 0 || width <= 0 || (height % bits) != 0) return SQLLM_E_SHAPE;
  sqllm_op op;
  memset(&op, 0, sizeof(op));
  op.bits = bits;
  op.K = height / bits * 32;  // height = K/32*bits (squeezellm/quant.py:48-51)
  op.N = width;
  op.batch = batch;
  if (batch > 0 && vec_height != op.K) return SQLLM_E_BATCH;
  op.vec = vec;
  op.qweight = mat;
  op.mul = mul;
  op.lookup_table = lut;
  if (has_csr) {
    if (num_rows != width) return SQLLM_E_SPARSE;  // CSR rows are output channels (quant.py:233)
    if (!rows) return SQLLM_E_NULL;
    op.rows = rows;
    op.cols = cols;
    op.vals = vals;
    op.nnz = nnz;
  }
  if (has_topx) {
    if (topX > 0 && !full_rows) return SQLLM_E_NULL;
    op.full_rows = topX > 0 ? full_rows : nullptr;
    op.full_row_indices = full_idx;
    op.topX = topX;
  }
  return sqllm_launch(&op, stream);
}

#define SQLLM_DENSE(BITS)                                                                          \
  int sqllm_vecquant##BITS##matmul_nuq_perchannel(const float* vec, const int32_t* mat, float* mul, \
                                                  const float* lookup_table, int height, int width, \
                                                  sqllm_stream_t stream) {                          \
    return named(BITS, 0, 0, vec, mat, mul, lookup_table, height, width, nullptr, nullptr, nullptr, \
                 0, 0, nullptr, nullptr, 0, false, false, stream);                                  \
  }                                                                                                 \
  int sqllm_vecquant##BITS##matmul_nuq_perchannel_batched(                                          \
      const float* vec, const int32_t* mat, float* mul, const float* lookup_table, int height,      \
      int width, int batch, int vec_height, sqllm_stream_t stream) {                                \
    if (batch < 1) return SQLLM_E_BATCH;                                                            \
    return named(BITS, batch, vec_height, vec, mat, mul, lookup_table, height, width, nullptr,      \
                 nullptr, nullptr, 0, 0, nullptr, nullptr, 0, false, false, stream);                \
  }

#define SQLLM_SPMV(BITS)                                                                            \
  int sqllm_vecquant##BITS##matmul_spmv_nuq_perchannel(                                             \
      const int32_t* rows, const int32_t* cols, const float* mat, const float* vec, float* mul,     \
      int num_rows, const int32_t* matq, const float* lookup_table, int height, int width, int nnz, \
      sqllm_stream_t stream) {                                                                      \
    return named(BITS, 0, 0, vec, matq, mul, lookup_table, height, width, rows, cols, mat, nnz,     \
                 num_rows, nullptr, nullptr, 0, true, false, stream);                               \
  }                                                                                                 \
  int sqllm_vecquant##BITS##matmul_spmv_nuq_perchannel_batched(                                     \
      const int32_t* rows, const int32_t* cols, const float* mat, const float* vec, float* mul,     \
      int num_rows, const int32_t* matq, const float* lookup_table, int height, int width, int nnz, \
      int batch, int vec_height, sqllm_stream_t stream) {                                           \
    if (batch < 1) return SQLLM_E_BATCH;                                                            \
    return named(BITS, batch, vec_height, vec, matq, mul, lookup_table, height, width, rows, cols,  \
                 mat, nnz, num_rows, nullptr, nullptr, 0, true, false, stream);                     \
  }                                                                                                 \
  int sqllm_vecquant##BITS##matmul_spmv_balanced_nuq_perchannel(                                    \
      const int32_t* rows, const int32_t* cols, const int32_t* startrows, const float* mat,         \
      const float* vec, float* mul, const int32_t* matq, const float* lookup_table, int num_rows,   \
      int num_threads, int numvals, int height, int width, sqllm_stream_t stream) {                 \
    (void)startrows;                                                                                \
    (void)num_threads;                                                                              \
    return named(BITS, 0, 0, vec, matq, mul, lookup_table, height, width, rows, cols, mat, numvals, \
                 num_rows, nullptr, nullptr, 0, true, false, stream);                               \
  }

#define SQLLM_HYBRID(BITS)                                                                          \
  int sqllm_vecquant##BITS##matmul_spmv_hybrid_nuq_perchannel(                                      \
      const int32_t* rows, const int32_t* cols, const float* mat, const float* vec,                 \
      const float* full_rows, const int32_t* full_row_indices, float* mul, int num_rows,            \
      const int32_t* matq, const float* lookup_table, int height, int width, int nnz, int topX,     \
      sqllm_stream_t stream) {                                                                      \
    return named(BITS, 0, 0, vec, matq, mul, lookup_table, height, width, rows, cols, mat, nnz,     \
                 num_rows, full_rows, full_row_indices, topX, true, true, stream);                  \
  }                                                                                                 \
  int sqllm_vecquant##BITS##matmul_spmv_hybrid_nuq_perchannel_batched(                              \
      const int32_t* rows, const int32_t* cols, const float* mat, const float* vec,                 \
      const float* full_rows, const int32_t* full_row_indices, float* mul, int num_rows,            \
      const int32_t* matq, const float* lookup_table, int height, int width, int nnz, int topX,     \
      int batch, int vec_height, sqllm_stream_t stream) {                                           \
    if (batch < 1) return SQLLM_E_BATCH;                                                            \
    return named(BITS, batch, vec_height, vec, matq, mul, lookup_table, height, width, rows, cols,  \
                 mat, nnz, num_rows, full_rows, full_row_indices, topX, true, true, stream);        \
  }

SQLLM_DENSE(3)
SQLLM_DENSE(4)
SQLLM_SPMV(3)
SQLLM_SPMV(4)
SQLLM_HYBRID(3)
SQLLM_HYBRID(4)

}  // extern "C"
