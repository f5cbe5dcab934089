// Internal interface between the C-ABI layer (sqllm_capi.hip) and the kernels (sqllm_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Measurement switches.  Every one of them selects a kernel variant that is slower or deliberately
// WRONG (an ablation); they exist only in measurement builds (python -m squeezellm_amd.build
// --ablation -> libsqllm_hip_ablation.so, never the product library).  A product build that sets
// one of them does not compile.
#ifndef SQLLM_ABLATION_BUILD
#if defined(SQLLM_PAIR3) || defined(SQLLM_PAIR3_NOCONFLICT) || defined(SQLLM_MFMA_VAR) || defined(SQLLM_MFMA_FAKE) || \
    defined(SQLLM_HALF_STAGES) || defined(SQLLM_WAVES) || defined(SQLLM_STREAM_RING) || defined(SQLLM_STREAM_WGCU) || \
    defined(SQLLM_STREAM_PRO) || defined(SQLLM_TOPX_ROWS) || defined(SQLLM_CSR_CHUNK)
#error "kernel variant switches need -DSQLLM_ABLATION_BUILD (python -m squeezellm_amd.build --ablation)"
#endif
#endif
#ifndef SQLLM_WAVES
#define SQLLM_WAVES 8  // waves per workgroup (measurement builds: 4)
#endif
#ifndef SQLLM_CSR_CHUNK
#define SQLLM_CSR_CHUNK 1024  // non-zeros per CSR workgroup (measurement builds: 2048)
#endif
#ifndef SQLLM_TOPX_ROWS
#define SQLLM_TOPX_ROWS 256  // k's per top-X slab (measured 128 / 256 / 512: profiles/r03_csr_chunk_topx_slab.txt)
#endif

namespace sqllm {

constexpr int kWaves = SQLLM_WAVES;          // waves per workgroup (512 threads)
constexpr int kTileN = 64;         // output columns per dense tile = 16 lanes x 4 (one dwordx4 each)
constexpr int kCsrChunk = SQLLM_CSR_CHUNK;  // non-zeros per CSR workgroup (a multiple of 1024: a lane holds a run of 2+)
constexpr int kCsrSpanMax = 2048;  // CSR rows a chunk may span and still accumulate in LDS
constexpr int kCsrXtSpan = 191;    // ... and still keep a 64-row tile of sums in LDS (wide batches, transposed vec; half of it: a 128-row tile)
constexpr int kSparsePassRows = 128;  // batch rows per workgroup of the wide-batch sparse launch
constexpr int kTopxRows = SQLLM_TOPX_ROWS;  // k's per top-X slab (a multiple of 32)
constexpr int kTopxLds = 1024;     // topX up to which slab sums are kept in LDS
constexpr int kMaxBatchTile = 8;   // batch rows handled per weight pass
constexpr int kMaxSlices = 120;    // K slices per column tile
constexpr int kMaxContrib = 63;    // fused linear: contributions one column may receive (6-bit count under the three non-finite flags)

// LDS floats of one kernel instantiation: max over roles of
//   dense: codebooks 4 column sub-tables * lut_entries * 32 slots (2 copies of 16 lanes) PLUS the
//          cross-wave slabs waves * BT * 64, the epilogue ticket and the fused linear's top-X
//          sums BT * 64 (the slabs must not overlap the codebooks: the combine is barrier-free)
//   csr  : kCsrSpanMax ints + kCsrSpanMax floats
//   topx : kTopxLds
constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int lds_floats(int lut_entries, int waves, int bt, bool pair3 = false) {
  // pair3: the 3-bit batch-1 kernels stage 64-entry PAIR tables, 4 x 64 x 128 B per tile
  return cmax((pair3 ? 4 * 64 * 32 : 4 * lut_entries * 32) + waves * bt * kTileN + 4 + bt * kTileN,
              cmax(2 * kCsrSpanMax, kTopxLds));
}

// Launch geometry, computed on the host (sqllm_capi.hip: make_plan) and passed by value.
struct KernelGeom {
  int K, N, batch;
  int col_tiles;        // ceil(N / 64)
  int units_total;      // K / 8 (w4: qweight rows) or K / 32 (w3: three-row units)
  int units_per_wg;     // K-slice length of a workgroup, in units (a multiple of waves * 4)
  int k_slices;         // ceil(units_total / units_per_wg)
  int dense_blocks;     // col_tiles * k_slices
  int dense_block0;     // first dense blockIdx.x (csr + topx blocks rounded up to a multiple of 8)
  int csr_blocks;       // ceil(nnz / kCsrChunk), 0 without a sparse term
  int topx_blocks;      // ceil(K / kTopxRows), 0 without a top-X term
  int nnz, topX;
  int sparse_last;      // 1: CSR / top-X workgroups come after the dense ones in the grid
  int fold_csr;         // 1: the CSR term is walked by the dense workgroups themselves (fused small launch: csr_tile_fold_staged; csr_blocks = 0)
  int csr_wide;         // 1: the CSR workgroups take 2 * kCsrChunk non-zeros each (csr_blocks counts those)
  int dense_prio;       // issue priority inside a batch-1 launch with sparse roles (sqllm_capi.hip: set_role_priority): 1 = the dense workgroups' waves at s_setprio 1, 2 = the CSR / top-X workgroups' waves, 0 = all equal
};

constexpr int kMaxSegments = 4;   // ops one launch can cover (they share vec, K, bits, batch)

// One op of a launch: its operands and its geometry.
struct Segment {
  const uint32_t* q;
  float* y;
  const float* lut;
  const int* rows;
  const int* cols;
  const float* vals;
  const float* full_rows;
  const int* full_idx;
  // fused-linear launches only (sqllm_linear_f16): `y` is then the plane of 64-bit accumulator
  // words [batch, N] in the caller's workspace (all zero between launches) and the finished
  // columns leave as fp16
  const float* bias;    // fp32 [N] or null
  void* out16;          // fp16 [batch, N]
  KernelGeom gm;
};

// Kernel argument block: up to kMaxSegments ops over the same input vector.  Workgroup ids
// [block0[s], block0[s+1]) belong to segment s (each range is a multiple of 8 long, so a dense
// tile's id modulo 8 -- its XCD -- does not depend on the segments before it).
struct GroupArgs {
  int n_seg;
  int block0[kMaxSegments + 1];
  Segment seg[kMaxSegments];
};

struct LaunchArgs {
  const void* x;        // fp32 (operator launches) or fp16 (fused-linear launches)
  bool linear = false;  // fused-linear launch: fp16 in / fp16 out, bias, self-cleaning workspace
  GroupArgs ga;
  hipEvent_t ev_start = nullptr;  // optional: recorded at this kernel's begin / end (profiling aid)
  hipEvent_t ev_stop = nullptr;
  int ablate = 0;  // measurement builds only (SQLLM_ABLATION_BUILD)
  int lds_pad = 0;  // measurement builds only: bytes of unused dynamic LDS per workgroup (caps the workgroups per CU)
  const float* xT = nullptr;  // wide-batch launches: transposed copy of x ([K, Bp]) for the CSR role, or null
  int Bp = 0;
  // wide-batch launches on the split matrix-core kernel: vec already split into bf16 planes (split_vec), or null
  const void* planes = nullptr;
  const uint32_t* plane_flags = nullptr;  // kSplitFlagWgs words: any non-zero lo part?
  bool wide = false;  // the geometry is make_plan_wide's: workgroups of eight column tiles (sqllm_mfma_wide.hip: sqllm_fused_wide)
  int wide_full_units = 0;  // ... and this many of them over all of K; the rest in gm.k_slices slices
  float* wide_slabs = nullptr;  // scratch for the slices' sums (wide_slab_bytes), or null: they add atomically
  int row_blocks = 0;  // tile form: blocks of 16 rows per pass (1, 2 or 4); 0 = mfma_row_blocks(batch)
};

// ---- streaming batch-1 kernel (sqllm_stream.hip) ----
constexpr int kStreamPieces4 = 2;  // 64-column tiles one workgroup's range may touch (codebook tables resident at once), 4-bit
constexpr int kStreamPieces3 = 2;  // ... 3-bit (32 KiB pair tables)

struct StreamSeg {  // one op of the launch, dense term only
  const uint32_t* q;
  float* y;
  const float* lut;
  int N;
  int tile0;  // index of the op's first tile in the launch's flattened tile space (INT_MAX: unused slot)
};

// Dense part of a streaming launch: the ops' 64-column tiles back to back, each tile `steps_per_tile`
// steps long (a step = 4 units = what one wave load covers), cut into ranges of `steps_per_wg` steps.
struct StreamArgs {
  const float* x;
  int K;
  int units_total;     // K / 8 (4-bit) or K / 32 (3-bit)
  int steps_per_tile;  // ceil(units_total / 4)
  int steps_per_wg;
  int total_steps;     // tiles of all ops * steps_per_tile
  int dense_block0;    // first dense workgroup id (the sparse-role workgroups come first)
  int n_dense;         // dense workgroups
  int n_seg;
  uint32_t s_magic;    // ceil(2^32 / steps_per_tile): tile of a step = mulhi(step, s_magic)
  StreamSeg seg[kMaxSegments];
  unsigned long long* probe;  // measurement builds: 8 timestamps per workgroup (tools/timeline.py); null otherwise
};

// batch rows handled per weight pass for a given batch size (template instantiations 1/2/4/8)
inline int batch_tile(int batch) { return batch <= 1 ? 1 : batch == 2 ? 2 : batch <= 4 ? 4 : 8; }
// ... of the fused batch-tile kernel behind the OPERATOR names (sqllm_fused_matvec<BITS, BT>, not the fused linear): a tile
// of exactly 3, 5, 6 or 7 rows too -- a 5-row batch on the 8-row tile pays the broadcasts and FMAs of 8 (round 6) -- and of the column-lane kernel
inline int batch_tile_op(int batch) { return (batch == 3 || batch == 5 || batch == 6 || batch == 7) ? batch : batch_tile(batch); }

// blocks of 16 batch rows one pass of the wide-batch (matrix-core) kernel covers: 1, 2 or 4
inline int mfma_row_blocks(int batch) { return batch <= 16 ? 1 : batch <= 32 ? 2 : 4; }

hipError_t launch_fused(int bits, const LaunchArgs& a, hipStream_t stream);
extern bool (*g_fused_variant)(int bits, const LaunchArgs& a, hipStream_t stream, hipError_t* err);  // measurement library hook (null in the product)
hipError_t launch_pair4(const LaunchArgs& a, hipStream_t stream);  // 4-bit, batch 1, operator launches: column-pair tables, 16-wave workgroups
// `ga`: the sparse roles of the launch (block0[] = prefix over csr + top-X workgroups only)
hipError_t launch_stream(int bits, const StreamArgs& sa, const GroupArgs& ga, hipStream_t stream, hipEvent_t e0, hipEvent_t e1,
                         int ablate);
hipError_t launch_batched_mfma(int bits, const LaunchArgs& a, hipStream_t stream);        // fp32 matrix instructions
hipError_t launch_batched_mfma_split(int bits, const LaunchArgs& a, hipStream_t stream);  // bf16 matrix instructions on exactly split operands (sqllm_mfma_split.hip)
hipError_t launch_batched_mfma_split_all(int bits, const LaunchArgs& a, hipStream_t stream);  // ... tile form, the op's sparse terms in the same grid
hipError_t launch_batched_mfma_wide(int bits, const LaunchArgs& a, hipStream_t stream);   // ... in the wide form (sqllm_mfma_wide.hip)
hipError_t launch_batched_cols(int bits, const LaunchArgs& a, hipStream_t stream);
constexpr int kSplitFlagWgs = 256;  // workgroups (and flag words) of split_vec
// planes of split_vec, in 16-byte chunks: 3 planes x 64 lanes per (16 rows, 32 k's); rows padded to a multiple of 64, every
// block of 16 rows K / 32 + 1 k blocks long, the last one all zero
inline uint64_t split_planes_chunks(int batch, int K) { return (uint64_t)((batch + 63) / 64 * 4) * (uint64_t)(K / 32 + 1) * 192u; }
// scratch for the sums of the wide form's K slices: 64 x 64 floats per slice and column tile
inline uint64_t wide_slab_bytes(int dense_blocks, int full_units, int k_slices) {
  return k_slices > 1 ? (uint64_t)(dense_blocks - full_units) * 8u * 64u * 64u * 4u : 0;
}
hipError_t split_vec(const float* x, void* planes, uint32_t* flags, int batch, int K, hipStream_t stream, hipEvent_t ev_start);
constexpr int kSmallSplitRows = 16;  // rows up to which a group of ops runs as ONE launch on the split matrix-core kernel (all three terms)
hipError_t launch_small_split(int bits, const LaunchArgs& a, hipStream_t stream);
hipError_t transpose_vec(const float* x, float* xT, int batch, int K, int Bp, hipStream_t stream, hipEvent_t ev_start);
// 2..16 rows: xT[k][rp], rp = the batch rounded up to a power of two (K * rp floats)
hipError_t transpose_small(const float* x, float* xT, int batch, int K, hipStream_t stream, hipEvent_t ev_start);
// ... + vec as three bf16 planes in fragment order, row block 0 (K / 32 + 1 k blocks of 3 KB; planes 16-byte aligned behind xT)
hipError_t prepare_small(const float* x, float* xT, void* planes, int batch, int K, hipStream_t stream, hipEvent_t ev_start);
inline int64_t small_planes_bytes(int K) { return (int64_t)(K / 32 + 1) * 3072; }
inline int64_t transpose_small_bytes(int batch, int K) { return (int64_t)K * (batch <= 2 ? 2 : batch <= 4 ? 4 : batch <= 8 ? 8 : 16) * 4; }
hipError_t launch_batched_sparse(const LaunchArgs& a, hipStream_t stream);
hipError_t check_csr(const int* rows, int N, int nnz, hipStream_t stream, int* bad);

}  // namespace sqllm
