// sqllm_pass.h -- device-side tables of the dependency-gated pass (sqllm_pass.hip): a whole decode pass --
// consecutive groups of ops, each group reading what the previous one wrote -- as ONE persistent launch.
// Shared between the planner (host) and the kernel; plain data, written once by sqllm_pass_build into the
// caller's workspace.
#pragma once
#include <stdint.h>

#include "sqllm_kernels.h"

namespace sqllm {

// A group's completion is a COUNT of its work items, kept in kPassShards words 64 bytes apart (one device-scope
// atomic per item; 256-1400 arrivals on ONE word would serialise at ~12 ns each -- MI355X_MICROARCH.md "fanin").
// Item i arrives at shard i % kPassShards; the pass knows how many each shard will receive.
constexpr int kPassShards = 8;
constexpr int kPassShardStride = 16;  // dwords between the shards of a group (64 B)
constexpr int kPassGroupStride = kPassShards * kPassShardStride;  // dwords of arrival state per group

enum PassRole : int { kPassDense = 0, kPassCsr = 1, kPassTopx = 2 };

// One op of the pass: 128 bytes, 128-byte aligned, in two halves of one 64-byte scalar load each -- what every item
// needs (and the dense items need nothing else), and what only the sparse items need.
struct PassSegHot {
  const uint32_t* q;
  float* y;
  const float* lut;
  const float* x;
  unsigned* arrive;  // arrival shards of this op's group; the shards of the group BEFORE it (the one whose completion
                     // this op's vec waits for) lie kPassGroupStride dwords below
  int K, N;
  int gate_group;    // index of the group this op's vec waits for (-1: none): a workgroup that has seen group g complete never polls for <= g again
  int gate_total;    // arrivals that group receives in all (its number of work items)
  int pad[2];
};
struct PassSegSparse {
  const int* rows;
  const int* cols;
  const float* vals;
  const float* full_rows;
  const int* full_idx;
  int nnz, topX;
  int col_tiles, units_total, units_per_wg;  // (the dense items carry their own geometry; kept for inspection)
  int group;         // index of this op's group
};
struct PassSeg {
  PassSegHot hot;
  PassSegSparse sp;
};
static_assert(sizeof(PassSegHot) == 64 && sizeof(PassSegSparse) == 64 && sizeof(PassSeg) == 128, "PassSeg is read with fixed-size scalar loads");

// One work item: a dense tile x K slice, a CSR chunk or a top-X slab of one op.  16 bytes: one scalar load.
struct PassItem {
  int seg_role;  // op index | role << 24
  int bid;       // dense: the tile's first column; CSR: chunk index; top-X: slab index
  int u_beg;     // dense: the K slice in units [u_beg, u_end)
  int u_end;
};

// status words in the workspace (zeroed before every launch together with the arrival shards)
enum PassStatus : int { kPassStatusError = 0, kPassStatusItem = 1, kPassStatusHead = 8, kPassStatusWords = 16 };  // [2..7]: diagnostics of a timeout

struct PassArgs {
  const PassItem* items;
  const PassSeg* segs;
  unsigned* status;
  int n_items;
  int poll_sleep;          // s_sleep(2) units between two polls of a gate
  unsigned timeout_ticks;  // 100 MHz ticks a gate may stay shut before the launch gives up (status[0] = 1)
  int pad;
  unsigned long long* timeline;  // measurement builds: 4 x u64 stamps per work item (tools/pass_timeline.py); null otherwise
};
static_assert(sizeof(PassArgs) == 48, "read with scalar loads");

// `device_args`: the PassArgs block inside the workspace image
hipError_t launch_pass(int bits, const PassArgs* device_args, int grid, hipStream_t stream, hipEvent_t e0, hipEvent_t e1);
hipError_t zero_pass_state(unsigned* state, int words, hipStream_t stream);
int pass_blocks_per_cu(int bits);  // resident workgroups per CU of the pass kernel (occupancy query; 0 on error)

}  // namespace sqllm
