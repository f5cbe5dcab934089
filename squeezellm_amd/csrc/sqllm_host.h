// sqllm_host.h -- internal interface of the host layer (sqllm_capi.hip): what the measurement library's code
// (csrc/experimental/) needs from it -- options, validation, the launch planner -- and the hooks through which
// that code takes part in option handling and launch routing.  The product library leaves every hook null.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

#include "sqllm_hip.h"
#include "sqllm_kernels.h"

namespace sqllm_host {

// Tuning knobs and debug switches are PER DEVICE (a set / get applies to the calling thread's current
// HIP device; slot 0 when no device is usable, e.g. in GPU-less planning tests): several GPUs driven
// from one process, or threads on different devices, do not steer each other's launches.
struct Knobs {
  std::atomic<int> target_wgs{0};
  std::atomic<int> groups_per_wave{0};
  std::atomic<int> cu_count{0};
  std::atomic<int> sparse_last{0};
  std::atomic<int> cols_groups{1};  // 0: a group of ops never takes the column-lane kernel (as before round 3)
  // Routing of the *_batched operators by batch size (0 = the measured defaults, which depend on the bit width:
  // 13B gate/up shape, profiles/r02_batch_paths_*.txt):
  //   4-bit: 2..4 rows column-lane kernel (or batch tiles: cols_pays), 5+ matrix cores (round 4; before: 5..8 batch tiles)
  //   3-bit: 2..8 rows column-lane kernel, 9+ matrix cores
  std::atomic<int> mfma_min_batch{0};  // rows from which the matrix-core kernel takes over
  std::atomic<int> cols_min_batch{0};  // the column-lane kernel serves cols_min_batch .. cols_max_batch rows (0 = default: 2)
  std::atomic<int> cols_max_batch{0};
  std::atomic<int> scratch_in_capture{1};  // stream-ordered scratch also while the stream is capturing (graph memory nodes)
  std::atomic<int> sparse_transpose{1};  // wide batches: the CSR role reads a transposed copy of vec (stream-ordered scratch)
  std::atomic<int> validate_csr{0};    // debug: check rows[] on the device before every launch that carries a CSR term
  std::atomic<int> mfma_split{1};      // wide batches: bf16 matrix instructions on exactly split operands (0: the fp32 matrix instruction)
  std::atomic<int> scratch_pool_threshold{1};  // 0: never touch the release threshold of the device's default memory pool
  std::atomic<int> mfma_wide_min_batch{0};  // rows from which a workgroup takes EIGHT column tiles, one per wave, all on the same k's (0 = the measured rule: when those units fill 80 % of the CUs; a huge value: never)
  std::atomic<int> split_planes_min_batch{0};  // rows from which vec is split ONCE into bf16 planes in scratch (0 = the measured default; a huge value: never)
  std::atomic<int> mfma_fuse_sparse{1};  // 17 rows up to the wide form: the op's CSR / top-X workgroups in the dense launch's grid (0: a launch of their own first)
  std::atomic<int> mfma_fuse_small{1};  // ... and up to 16 rows: the group's ops with their sparse terms as ONE launch of that kernel
  std::atomic<int> small_reserve_topx{0};  // fused small launch: 1 = plan the dense ranges for the slots the top-X slabs leave (measured: the coarser ranges cost more than the late starters, profiles/r05_small_split_reserve.txt)
  std::atomic<int> small_planes{1};  // fused small launch with a transposed vec: its dense term loads vec already split into bf16 planes (written by the same kernel in front) instead of splitting in registers
  std::atomic<int> small_wgs_per_cu{0};  // fused small launch of the split kernel: dense workgroups per CU its planner aims at (0 = default)
};
constexpr int kMaxDevices = 32;

int device_slot();  // index of the calling thread's current device in per-device tables (0 when none is usable)
Knobs& knobs();
int cu_count();
int validate(const sqllm_op* op);
int validate_csr_values(const sqllm_op* op, sqllm_stream_t stream);
void make_plan(const sqllm_op* op, sqllm::KernelGeom* gm, int ops_in_launch = 1, int max_slices = sqllm::kMaxSlices,
               int waves = sqllm::kWaves);
void fill_segment(const sqllm_op* op, sqllm::Segment* sg);  // the operands of an op as a launch segment (geometry not touched)

// Hooks of the measurement library (python -m squeezellm_amd.build --ablation; csrc/experimental/sqllm_experimental.hip
// installs them from a static initialiser).  All null in the product library.
struct ExperimentalHooks {
  int (*set_option)(const char* name, int value) = nullptr;  // SQLLM_E_OPTION: not one of mine
  int (*get_option)(const char* name, int* value) = nullptr;
  // a group of operator ops (not fused linears) about to be launched: true = the hook has taken the launch, *rc is its result
  bool (*route)(const sqllm_op* ops, int n, sqllm_stream_t stream, hipEvent_t e0, hipEvent_t e1, int* rc) = nullptr;
  // last word on a product launch's arguments: ablation bits, unused-LDS pad, timeline buffer
  void (*decorate)(sqllm::LaunchArgs* a) = nullptr;
  int (*csr_ablation_bits)() = nullptr;  // ride along in KernelGeom::sparse_last (bits 1..)
  // timing experiment: the fused small launch WITHOUT the kernel in front of it (it then reads whatever the workspace
  // holds: results are garbage by design -- which is why the product library has no such mode)
  bool (*skip_prepare_small)() = nullptr;
};
extern ExperimentalHooks g_experimental;

}  // namespace sqllm_host
