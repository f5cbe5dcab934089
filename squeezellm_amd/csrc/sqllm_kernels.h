// Internal interface between the C-ABI layer (sqllm_capi.hip) and the kernels (sqllm_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sqllm {

constexpr int kThreads = 256;      // workgroup size: 4 wave64
constexpr int kWaves = 4;
constexpr int kTileN = 256;        // output columns per dense tile = 64 lanes x 4 (one dwordx4 each)
constexpr int kCsrChunk = 1024;    // non-zeros per CSR workgroup
constexpr int kCsrSpanMax = 2048;  // CSR rows a chunk may span and still accumulate in LDS
constexpr int kTopxRows = 128;     // k's per top-X slab
constexpr int kTopxLds = 1024;     // topX up to which slab sums are kept in LDS
constexpr int kMaxBatchTile = 8;   // batch rows handled per weight pass

// LDS floats: max over roles of
//   dense: codebooks 4 * 16 * 64 = 4096 (w4) ; cross-wave reduction kWaves * BT * 256 <= 8192
//   csr  : kCsrSpanMax ints + kCsrSpanMax floats = 4096
//   topx : kTopxLds
constexpr int kLdsFloats = kWaves * kMaxBatchTile * kTileN;

// Launch geometry, computed on the host (sqllm_capi.hip: make_plan) and passed by value.
struct KernelGeom {
  int K, N, batch;
  int col_tiles;        // ceil(N / 256)
  int groups_total;     // K / 8 (w4 rows) or K / 32 (w3 three-row groups)
  int groups_per_wave;  // groups each wave walks; a workgroup covers 4x that
  int k_slices;         // ceil(groups_total / (4 * groups_per_wave))
  int dense_blocks;     // col_tiles * k_slices
  int dense_block0;     // first dense blockIdx.x (csr + topx blocks rounded up to a multiple of 8)
  int csr_blocks;       // ceil(nnz / kCsrChunk), 0 without a sparse term
  int topx_blocks;      // ceil(K / kTopxRows), 0 without a top-X term
  int nnz, topX;
};

struct LaunchArgs {
  const float* x;
  const uint32_t* q;
  float* y;
  const float* lut;
  const int* rows;
  const int* cols;
  const float* vals;
  const float* full_rows;
  const int* full_idx;
  KernelGeom gm;
};

// batch rows handled per weight pass for a given batch size (template instantiations 1/2/4/8)
inline int batch_tile(int batch) { return batch <= 1 ? 1 : batch == 2 ? 2 : batch <= 4 ? 4 : 8; }

hipError_t launch_fused(int bits, const LaunchArgs& a, hipStream_t stream);

}  // namespace sqllm
