// sqllm_capi.hip -- the extern "C" surface declared in include/sqllm_hip.h: argument validation,
// launch planning, and the reference operator names as thin adapters over sqllm_launch().
// Replaces the reference's pybind11 layer squeezellm/quant_cuda.cpp:112-270 and the grid math of
// its launchers squeezellm/quant_cuda_kernel.cu:132-738 (no torch types cross this boundary).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <atomic>

#include <vector>

#include "sqllm_hip.h"
#include "sqllm_kernels.h"
#include "sqllm_pass.h"

namespace {

// Tuning knobs and debug switches are PER DEVICE (a set / get applies to the calling thread's current
// HIP device; slot 0 when no device is usable, e.g. in GPU-less planning tests): several GPUs driven
// from one process, or threads on different devices, do not steer each other's launches.
struct Knobs {
  std::atomic<int> target_wgs{0};
  std::atomic<int> groups_per_wave{0};
  std::atomic<int> cu_count{0};
  std::atomic<int> ablate{0};
  std::atomic<int> lds_pad{0};  // measurement builds: unused dynamic LDS per workgroup, bytes
  std::atomic<int> sparse_last{0};
  std::atomic<int> cols_groups{1};  // 0: a group of ops never takes the column-lane kernel (as before round 3)
  std::atomic<int> ablate_csr{0};
  // Routing of the *_batched operators by batch size (0 = the measured defaults, which depend on the bit width:
  // 13B gate/up shape, profiles/r02_batch_paths_*.txt):
  //   4-bit: 2..4 rows column-lane kernel, 5..8 batch tiles of the batch-1 kernel, 9+ matrix cores
  //   3-bit: 2..16 rows column-lane kernel (two passes from 9 rows), 17+ matrix cores
  std::atomic<int> mfma_min_batch{0};  // rows from which the matrix-core kernel takes over
  std::atomic<int> cols_min_batch{0};  // the column-lane kernel serves cols_min_batch .. cols_max_batch rows (0 = default: 2)
  std::atomic<int> cols_max_batch{0};
  std::atomic<int> scratch_in_capture{1};  // stream-ordered scratch also while the stream is capturing (graph memory nodes)
  std::atomic<int> sparse_transpose{1};  // wide batches: the CSR role reads a transposed copy of vec (stream-ordered scratch)
  std::atomic<int> validate_csr{0};    // debug: check rows[] on the device before every launch that carries a CSR term
  std::atomic<int> pair4{-1};          // 4-bit batch-1 operator launches on the column-pair-table kernel (sqllm_pair.hip): -1 default, 0 / 1
  std::atomic<int> pair4_min_mb{12};   // ... from this many MB of packed weights per launch (below it the fused kernel's small tables win)
  std::atomic<int> stream{-1};         // batch-1 operator launches on the streaming kernel: -1 = default (off), 0 / 1
  std::atomic<void*> timeline{nullptr};  // measurement build: per-workgroup timestamp buffer
  // dependency-gated pass (sqllm_pass.hip)
  std::atomic<int> pass_poll_sleep{4};     // s_sleep(2) units between two polls of a gate
  std::atomic<int> pass_timeout_ms{2000};  // a gate that stays shut this long ends the launch with status 1
  std::atomic<int> pass_wgs_per_cu{0};     // resident workgroups per CU (0 = the occupancy query's answer)
};
constexpr int kMaxDevices = 32;
Knobs g_knobs[kMaxDevices];

Knobs& knobs() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();  // no usable device: not an error of the call being served
    dev = 0;
  }
  if (dev < 0 || dev >= kMaxDevices) dev = 0;
  return g_knobs[dev];
}

// The wide-batch CSR path takes stream-ordered scratch (hipMallocAsync) per op.  The default pool's
// release threshold is 0: every synchronisation hands the block back to the OS and the next call pays
// a real allocation + map.  Raise it once per device (never lower it) so that the pool keeps what one
// op needs (2048 rows x K = 22016 floats is 180 MB).
void keep_scratch_in_pool() {
  static std::atomic<unsigned> done{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) { (void)hipGetLastError(); return; }
  const unsigned bit = 1u << dev;
  if (done.load(std::memory_order_relaxed) & bit) return;
  done.fetch_or(bit, std::memory_order_relaxed);
  hipMemPool_t pool = nullptr;
  if (hipDeviceGetDefaultMemPool(&pool, dev) != hipSuccess || !pool) { (void)hipGetLastError(); return; }
  uint64_t cur = 0;
  const uint64_t want = 256ull << 20;
  if (hipMemPoolGetAttribute(pool, hipMemPoolAttrReleaseThreshold, &cur) != hipSuccess) { (void)hipGetLastError(); cur = 0; }
  if (cur < want) {
    uint64_t v = want;
    if (hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &v) != hipSuccess) (void)hipGetLastError();
  }
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
void make_plan(const sqllm_op* op, sqllm::KernelGeom* gm, int ops_in_launch = 1, int max_slices = sqllm::kMaxSlices,
               int waves = sqllm::kWaves) {
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
      target = (mb <= 12.0 ? (alone ? 2 : 1) : (!alone && mb <= 16.0) ? 2 : (alone ? 4 : 3)) * cu_count();
      if (waves > sqllm::kWaves) target = target * sqllm::kWaves / waves;  // (16-wave workgroups: half as many, twice the rows each)
    }
    int slices = (target + gm->col_tiles / 2) / gm->col_tiles;
    if (slices < 1) slices = 1;
    if (slices > max_slices) slices = max_slices;
    upw = (gm->units_total + slices - 1) / slices;
  }
  upw = (upw + step - 1) / step * step;
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
#ifdef SQLLM_ABLATION_BUILD
  gm->sparse_last |= knobs().ablate_csr.load(std::memory_order_relaxed) << 1;  // CSR-role ablation bits ride along
#endif
}

// Geometry of the wide-batch (matrix-core) kernel: one pass covers 16 * mb batch rows (blockIdx.y
// walks the passes).  The dense work of a pass is the FLATTENED (column tile, unit) space cut into
// equal contiguous ranges, one per workgroup, as many as the chip holds at once (2 per CU; 1 for
// the 64-row kernels, by their registers) divided by the number of passes: one round of
// workgroups, none of them short (N / 64 is rarely a multiple of the CU count: cutting K slices per
// column tile left the last round 27 % full on the 13B gate/up shape).  A range is a whole number
// of workgroup steps (waves x 4 units); one that crosses a tile boundary costs a second piece.
void make_plan_mfma(const sqllm_op* op, sqllm::KernelGeom* gm) {
  make_plan(op, gm, 1);
  const int mb = sqllm::mfma_row_blocks(gm->batch);
  const int grid_y = (gm->batch + 16 * mb - 1) / (16 * mb);
  const int step = sqllm::kWaves * 4;
  const long long total_units = (long long)gm->col_tiles * gm->units_total;
  long long upw = (long long)knobs().groups_per_wave.load(std::memory_order_relaxed) * step;
  bool aligned = false;
  if (upw <= 0) {
    int target = knobs().target_wgs.load(std::memory_order_relaxed);
    if (target <= 0) target = (mb == 4 ? 1 : 2) * cu_count();
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
  }
  upw = (upw + step - 1) / step * step;
  if (upw > 0x3fffffff) upw = 0x3fffffff / step * step;
  gm->units_per_wg = (int)upw;
  gm->k_slices = (int)((gm->units_total + upw - 1) / upw);  // pieces per column tile (reported by plan_query)
  gm->dense_blocks = aligned ? gm->col_tiles * gm->k_slices : (int)((total_units + upw - 1) / upw);
  gm->sparse_last = 0;
  gm->dense_block0 = (gm->csr_blocks + gm->topx_blocks + 7) / 8 * 8;
}

int mfma_min_batch_of(const sqllm_op* op) {
  const int v = knobs().mfma_min_batch.load(std::memory_order_relaxed);
  return v > 0 ? v : (op->bits == 3 ? 17 : 9);
}
int cols_max_batch_of(const sqllm_op* op) {
  const int v = knobs().cols_max_batch.load(std::memory_order_relaxed);
  return v > 0 ? v : (op->bits == 3 ? 16 : 4);
}

bool takes_mfma_path(const sqllm_op* op) { return op->batch >= 1 && op->batch >= mfma_min_batch_of(op); }

// Geometry of the small-batch column-lane kernel: passes of batch_tile(batch) <= 8 rows
// (blockIdx.y); the dense work of a pass is cut into equal ranges of the flattened
// (column tile, unit) space like make_plan_mfma's, three workgroups per CU (the phases of a
// workgroup -- table build, decode, combine -- hide behind its neighbours').
void make_plan_cols(const sqllm_op* op, sqllm::KernelGeom* gm, int ops_in_launch = 1) {
  make_plan(op, gm, 1);
  const int bt = sqllm::batch_tile(gm->batch);
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
  }
  upw = (upw + sqllm::kWaves - 1) / sqllm::kWaves * sqllm::kWaves;
  if (upw > 0x3fffffff) upw = 0x3fffffff / sqllm::kWaves * sqllm::kWaves;
  gm->units_per_wg = (int)upw;
  gm->k_slices = (int)((gm->units_total + upw - 1) / upw);
  // (the kernel recognises the tile-aligned cut by dense_blocks == col_tiles * k_slices -- which, when it holds
  // for a contiguous cut too, describes the same ranges)
  gm->dense_blocks = aligned ? gm->col_tiles * gm->k_slices : (int)((total_units + upw - 1) / upw);
  gm->sparse_last = 0;
  gm->dense_block0 = (gm->csr_blocks + gm->topx_blocks + 7) / 8 * 8;
}

int cols_min_batch_of() {
  const int v = knobs().cols_min_batch.load(std::memory_order_relaxed);
  return v > 0 ? v : 2;
}

// Column-pair-table kernel (sqllm_pair.hip): 4-bit operator launches at batch 1 whose packed weights are large
// enough to pay for the 64 KiB tables (option pair4_min_mb, MB per launch).
bool takes_pair4_path(const sqllm_op* ops, int n) {
#ifndef SQLLM_ABLATION_BUILD
  (void)ops; (void)n;
  return false;
#endif
  const int v = knobs().pair4.load(std::memory_order_relaxed);
  if (v == 0 || ops[0].bits != 4) return false;
  if (v < 0) return false;  // default: off until measured
  double mb = 0.0;
  for (int i = 0; i < n; ++i) {
    if (ops[i].batch > 1 || (ops[i].N % 4) != 0) return false;
    mb += (double)ops[i].K * ops[i].N / 2.0 / 1e6;
  }
  return mb >= (double)knobs().pair4_min_mb.load(std::memory_order_relaxed);
}

// Streaming batch-1 kernel (sqllm_stream.hip): does this launch take it, and with what geometry?
bool takes_stream_path(const sqllm_op* ops, int n) {
#ifndef SQLLM_ABLATION_BUILD
  (void)ops; (void)n;
  return false;
#endif
  const int v = knobs().stream.load(std::memory_order_relaxed);
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
  if (op->bits == 4) return n_ops >= 3 || (n_ops == 1 && mb >= 20.0);
  if (op->batch <= 4 && mb >= 16.0) return true;
  return op->N >= 8192;
}

// a group of ops over one vec (q/k/v, gate/up) is judged as the one op it is to the kernel: the sum of its columns
bool group_takes_cols_path(const sqllm_op* ops, int n) {
  sqllm_op sum = ops[0];
  long long N = 0;
  for (int i = 0; i < n; ++i) N += ops[i].N;
  sum.N = N > 0x7fffffff ? 0x7fffffff : (int)N;
  return !takes_mfma_path(&ops[0]) && sum.batch >= 1 && sum.batch >= cols_min_batch_of() && sum.batch <= cols_max_batch_of(&sum) && cols_pays(&sum, n);
}

bool takes_cols_path(const sqllm_op* op) {
  return !takes_mfma_path(op) && op->batch >= 1 && op->batch >= cols_min_batch_of() && op->batch <= cols_max_batch_of(op) && cols_pays(op, 1);
}

}  // namespace

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
    case SQLLM_E_WORKSPACE: return "pass workspace too small, misaligned, or not the one the pass was built for";
    default: break;
  }
  if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
  return "unknown sqllm error";
}

#ifdef SQLLM_ABLATION_BUILD
// measurement build only (not in the header): device buffer of 8 x u64 per workgroup of the next launches
void sqllm_debug_set_timeline(void* buf) { knobs().timeline.store(buf); }
// measurement build only: the streaming kernel's plan for a group of ops -> {takes_stream, n_dense, steps_per_wg, steps_per_tile, total_steps}
void sqllm_debug_stream_plan(const sqllm_op* ops, int n, int sparse_blocks, int* out) {
  sqllm::StreamArgs sa;
  out[0] = takes_stream_path(ops, n) ? 1 : 0;
  make_plan_stream(ops, n, sparse_blocks, &sa);
  out[1] = sa.n_dense; out[2] = sa.steps_per_wg; out[3] = sa.steps_per_tile; out[4] = sa.total_steps;
}
#endif

int sqllm_set_option(const char* name, int value) {
  if (!name || value < 0) return SQLLM_E_OPTION;
  if (!strcmp(name, "target_wgs")) { knobs().target_wgs.store(value); return SQLLM_OK; }
  if (!strcmp(name, "groups_per_wave")) { knobs().groups_per_wave.store(value); return SQLLM_OK; }
  if (!strcmp(name, "sparse_last")) { knobs().sparse_last.store(value ? 1 : 0); return SQLLM_OK; }
  if (!strcmp(name, "cols_groups")) { knobs().cols_groups.store(value ? 1 : 0); return SQLLM_OK; }
  if (!strcmp(name, "cu_count")) { knobs().cu_count.store(value); return SQLLM_OK; }  // for GPU-less planning tests
  if (!strcmp(name, "mfma_min_batch")) { knobs().mfma_min_batch.store(value); return SQLLM_OK; }
  if (!strcmp(name, "cols_min_batch")) { knobs().cols_min_batch.store(value); return SQLLM_OK; }
  if (!strcmp(name, "cols_max_batch")) { knobs().cols_max_batch.store(value); return SQLLM_OK; }
  if (!strcmp(name, "sparse_transpose")) { knobs().sparse_transpose.store(value ? 1 : 0); return SQLLM_OK; }
  if (!strcmp(name, "scratch_in_capture")) { knobs().scratch_in_capture.store(value ? 1 : 0); return SQLLM_OK; }
  if (!strcmp(name, "validate_csr")) { knobs().validate_csr.store(value ? 1 : 0); return SQLLM_OK; }
  if (!strcmp(name, "pass_poll_sleep")) { knobs().pass_poll_sleep.store(value); return SQLLM_OK; }
  if (!strcmp(name, "pass_timeout_ms")) { knobs().pass_timeout_ms.store(value > 0 ? value : 1); return SQLLM_OK; }
  if (!strcmp(name, "pass_wgs_per_cu")) { knobs().pass_wgs_per_cu.store(value); return SQLLM_OK; }
#ifdef SQLLM_ABLATION_BUILD
  // the measured-and-not-adopted kernels (sqllm_stream.hip, sqllm_pair.hip) exist in the measurement library only
  if (!strcmp(name, "stream")) { knobs().stream.store(value > 1 ? -1 : value); return SQLLM_OK; }  // 0 off, 1 on, 2 default (off)
  if (!strcmp(name, "pair4")) { knobs().pair4.store(value > 1 ? -1 : value); return SQLLM_OK; }    // 0 off, 1 on, 2 default (off)
  if (!strcmp(name, "pair4_min_mb")) { knobs().pair4_min_mb.store(value); return SQLLM_OK; }
  if (!strcmp(name, "ablate")) { knobs().ablate.store(value); return SQLLM_OK; }
  if (!strcmp(name, "lds_pad")) { knobs().lds_pad.store(value); return SQLLM_OK; }
  if (!strcmp(name, "ablate_csr")) { knobs().ablate_csr.store(value); return SQLLM_OK; }
#endif
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
  if (!strcmp(name, "pass_poll_sleep")) { *value = knobs().pass_poll_sleep.load(); return SQLLM_OK; }
  if (!strcmp(name, "pass_timeout_ms")) { *value = knobs().pass_timeout_ms.load(); return SQLLM_OK; }
  if (!strcmp(name, "pass_wgs_per_cu")) { *value = knobs().pass_wgs_per_cu.load(); return SQLLM_OK; }
#ifdef SQLLM_ABLATION_BUILD
  if (!strcmp(name, "stream")) { const int v = knobs().stream.load(); *value = v < 0 ? 2 : v; return SQLLM_OK; }
  if (!strcmp(name, "pair4")) { const int v = knobs().pair4.load(); *value = v < 0 ? 2 : v; return SQLLM_OK; }
  if (!strcmp(name, "pair4_min_mb")) { *value = knobs().pair4_min_mb.load(); return SQLLM_OK; }
#endif
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
  if (mfma) make_plan_mfma(op, &gm);
  else if (takes_cols_path(op)) make_plan_cols(op, &gm);
  else make_plan(op, &gm);
  plan->col_tiles = gm.col_tiles;
  plan->k_slices = gm.k_slices;
  plan->groups_per_wave = gm.units_per_wg;
  plan->dense_blocks = gm.dense_blocks;
  plan->csr_blocks = gm.csr_blocks;
  plan->topx_blocks = gm.topx_blocks;
  plan->grid_x = mfma ? gm.dense_blocks : gm.dense_block0 + gm.dense_blocks;  // (wide batches: the sparse terms are a launch of their own)
  const int rows_per_pass = mfma ? 16 * sqllm::mfma_row_blocks(gm.batch) : sqllm::batch_tile(gm.batch);
  plan->grid_y = (gm.batch + rows_per_pass - 1) / rows_per_pass;
  return SQLLM_OK;
}

static int64_t align16(int64_t v) { return (v + 15) / 16 * 16; }

// workspace of a fused linear: 64-bit accumulator words [batch, N]
int64_t sqllm_linear_workspace_bytes(const sqllm_op* op) {
  if (!op || op->N <= 0) return 0;
  return align16(8ll * (op->batch <= 0 ? 1 : op->batch) * op->N);
}

// One kernel over 1..kMaxSegments ops that share vec, K, bits and batch.  `lin` (optional) points at
// the fused-linear descriptors the ops were taken from: `ops` is then lin[i].op.
static int launch_group_with_events(const sqllm_op* ops, int n, sqllm_stream_t stream, hipEvent_t e0,
                                    hipEvent_t e1, const sqllm_linear* lin = nullptr) {
  if (n < 1 || n > sqllm::kMaxSegments) return SQLLM_E_GROUP;
  if (!ops && !lin) return SQLLM_E_NULL;
  // (the column-lane kernel only for an op that is alone in its launch: q/k/v or gate/up sharing one
  // launch of the batch tiles beat three / two launches of it -- 13B s45 decoder layer at 2 rows:
  // 88 vs 101 us)
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
  if (!lin && (takes_mfma_path(&ops[0]) || (n == 1 && takes_cols_path(&ops[0])))) {
    // batched operators: one launch per op (the members of a group only share their input) of the
    // matrix-core kernel (wide batches) or of the column-lane kernel (small ones)
    const bool mfma = takes_mfma_path(&ops[0]);
    // Wide batches with a CSR term: the role wants vec TRANSPOSED (lane = batch row: one coalesced
    // read per non-zero instead of `batch` gathers).  The copy lives in stream-ordered scratch
    // (hipMallocAsync / hipFreeAsync on the caller's stream: no host synchronisation, the pool keeps
    // the block for the next call).  Without scratch the role falls back to gathering from vec itself.
    float* xT = nullptr;
    int Bp = 0;
    bool any_csr = false;
    for (int i = 0; i < n; ++i) any_csr = any_csr || (ops[i].nnz > 0 && ops[i].rows && ops[i].cols && ops[i].vals);
    if (mfma && any_csr && ops[0].vec && ops[0].batch > 0 && ops[0].K > 0 && knobs().sparse_transpose.load(std::memory_order_relaxed)) {
      // (inside a stream capture the allocation and the free become memory nodes of the graph -- works under
      // torch's graph capture on ROCm 7.2; option scratch_in_capture = 0 keeps captures allocation-free)
      bool scratch_ok = true;
      if (!knobs().scratch_in_capture.load(std::memory_order_relaxed)) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        scratch_ok = hipStreamIsCapturing(static_cast<hipStream_t>(stream), &cs) == hipSuccess && cs == hipStreamCaptureStatusNone;
      }
      if (scratch_ok) {
        Bp = (ops[0].batch + 63) / 64 * 64;
        keep_scratch_in_pool();
        void* p = nullptr;
        if (hipMallocAsync(&p, (size_t)ops[0].K * Bp * sizeof(float), static_cast<hipStream_t>(stream)) == hipSuccess && p) {
          xT = static_cast<float*>(p);
          hipError_t e = sqllm::transpose_vec(ops[0].vec, xT, ops[0].batch, ops[0].K, Bp, static_cast<hipStream_t>(stream), e0);
          if (e == hipSuccess) e0 = nullptr;  // (a profiled group starts with its transpose)
          if (e != hipSuccess) { (void)hipFreeAsync(p, static_cast<hipStream_t>(stream)); return static_cast<int>(e); }
        } else {
          (void)hipGetLastError();  // no scratch: gather from vec
          Bp = 0;
        }
      } else {
        (void)hipGetLastError();
      }
    }
    struct ScratchGuard {
      float* p; hipStream_t s;
      ~ScratchGuard() { if (p) (void)hipFreeAsync(p, s); }
    } guard{xT, static_cast<hipStream_t>(stream)};
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
      if (mfma) make_plan_mfma(op, &sg.gm);
      else make_plan_cols(op, &sg.gm);
      a.ga.block0[0] = 0;
      for (int j = 1; j <= sqllm::kMaxSegments; ++j) a.ga.block0[j] = sg.gm.dense_block0 + sg.gm.dense_blocks;
      if (mfma) {
        // the sparse terms first, as a launch of their own (see sqllm_sparse_batched), then the dense term
        // (running the two on different streams was tried: they do not overlap -- the dense kernel holds
        // every CU's registers -- and the two event waits cost 14 us per op)
        const bool sparse = sg.gm.csr_blocks + sg.gm.topx_blocks > 0;
        if (sparse) {
          sqllm::LaunchArgs as = a;
          as.ev_stop = nullptr;
#ifdef SQLLM_ABLATION_BUILD
          as.ga.seg[0].bias = static_cast<const float*>(knobs().timeline.load(std::memory_order_relaxed));  // timeline probe (tools/timeline.py --batch)
#endif
          rc = static_cast<int>(sqllm::launch_batched_sparse(as, static_cast<hipStream_t>(stream)));
          if (rc != SQLLM_OK) return rc;
          a.ev_start = nullptr;
        }
        rc = static_cast<int>(sqllm::launch_batched_mfma(op->bits, a, static_cast<hipStream_t>(stream)));
      } else {
        rc = static_cast<int>(sqllm::launch_batched_cols(op->bits, a, static_cast<hipStream_t>(stream)));
      }
      if (rc != SQLLM_OK) return rc;
    }
    return SQLLM_OK;
  }
  if (!lin && takes_stream_path(ops, n)) {
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
      sg.q = reinterpret_cast<const uint32_t*>(op->qweight);
      sg.y = op->mul;
      sg.lut = op->lookup_table;
      sg.rows = op->rows;
      sg.cols = op->cols;
      sg.vals = op->vals;
      sg.full_rows = op->topX > 0 ? op->full_rows : nullptr;
      sg.full_idx = op->topX > 0 ? op->full_row_indices : nullptr;
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
#ifdef SQLLM_ABLATION_BUILD
    sa.probe = static_cast<unsigned long long*>(knobs().timeline.load(std::memory_order_relaxed));
#endif
#ifdef SQLLM_ABLATION_BUILD
    return static_cast<int>(sqllm::launch_stream(ops[0].bits, sa, ga, static_cast<hipStream_t>(stream), e0, e1,
                                                 knobs().ablate.load(std::memory_order_relaxed)));
#endif
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
  const bool pair4 = !lin && takes_pair4_path(ops, n);
  sqllm::LaunchArgs a;
  a.linear = lin != nullptr;
  a.ev_start = e0;
  a.ev_stop = e1;
  a.ablate = knobs().ablate.load(std::memory_order_relaxed);
  a.lds_pad = knobs().lds_pad.load(std::memory_order_relaxed);
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
    sg.q = reinterpret_cast<const uint32_t*>(op->qweight);
    sg.y = op->mul;
    sg.lut = op->lookup_table;
    sg.rows = op->rows;
    sg.cols = op->cols;
    sg.vals = op->vals;
    // (full_rows without columns is no term at all: the kernels key the top-X work on the pointer)
    sg.full_rows = op->topX > 0 ? op->full_rows : nullptr;
    sg.full_idx = op->topX > 0 ? op->full_row_indices : nullptr;
    sg.bias = nullptr;
    sg.out16 = nullptr;
#ifdef SQLLM_ABLATION_BUILD
    if (!lin) sg.bias = static_cast<const float*>(knobs().timeline.load(std::memory_order_relaxed));
#endif
    // fused linear: a column's K slices + the CSR chunks its row can be spread over must fit the 6-bit count
    const int csr_bound = (op->rows && op->nnz > 0) ? op->K / sqllm::kCsrChunk + 2 : 0;
    make_plan(op, &sg.gm, n, lin ? (sqllm::kMaxContrib - csr_bound > 1 ? sqllm::kMaxContrib - csr_bound : 1) : sqllm::kMaxSlices,
              pair4 ? 16 : sqllm::kWaves);
    if (pair4) sg.gm.sparse_last = 0;
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
#ifdef SQLLM_ABLATION_BUILD
  if (pair4) return static_cast<int>(sqllm::launch_pair4(a, static_cast<hipStream_t>(stream)));
#endif
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
      rc = launch_group_with_events(ops + at, group_sizes[g], stream, ev[2 * g], ev[2 * g + 1]);
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

// ---- dependency-gated pass (sqllm_pass.hip) ------------------------------------------------------
// Workspace image:  [status words | arrival shards, kPassGroupStride dwords per group]  <- zeroed before every launch
//                   [PassArgs, 128 bytes] [PassSeg per op, 128-byte aligned] [PassItem per work item]
namespace {

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
    a->poll_sleep = knobs().pass_poll_sleep.load(std::memory_order_relaxed);
    const long long ticks = (long long)knobs().pass_timeout_ms.load(std::memory_order_relaxed) * 100000ll;  // 100 MHz
    a->timeout_ticks = ticks > 0xffffffffll ? 0xffffffffu : (unsigned)ticks;
#ifdef SQLLM_ABLATION_BUILD
    a->timeline = static_cast<unsigned long long*>(knobs().timeline.load(std::memory_order_relaxed));
#endif
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
  int per_cu = knobs().pass_wgs_per_cu.load(std::memory_order_relaxed);
  if (per_cu <= 0) {
    per_cu = sqllm::pass_blocks_per_cu(bits);
    if (per_cu <= 0) per_cu = 4;  // (no device to ask: the kernel is built for four per CU -- tests/test_codegen_cpu.py)
  }
  // (work items are taken from a queue: the grid only has to be what the chip CAN hold, not what it WILL)
  long long grid = (long long)per_cu * cu_count();
  if (grid > lay.n_items) grid = lay.n_items;
  pass->grid = (int32_t)grid;
  pass->poll_sleep = knobs().pass_poll_sleep.load(std::memory_order_relaxed);
  pass->timeout_ms = knobs().pass_timeout_ms.load(std::memory_order_relaxed);
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
