/* sqllm_pass_api.h -- C ABI of the dependency-gated pass.  MEASUREMENT LIBRARY ONLY (libsqllm_hip_ablation.so):
 * built, parity-green, measured on MI355X and NOT adopted -- 2.3-4x slower than one launch per group
 * (profiles/r04_pass_*.txt).  Kept so that the measurement stays reproducible. */
#ifndef SQLLM_PASS_API_H
#define SQLLM_PASS_API_H

#include <stdint.h>

#include "sqllm_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SQLLM_E_WORKSPACE (-9) /* pass workspace too small / misaligned / not the one the pass was built for */

/* ---------------------------------------------------------------------------------------------
 * Dependency-gated pass: consecutive groups of batch-1 ops as ONE persistent launch.
 *
 * sqllm_launch_groups enqueues one kernel per group; the groups of a decode pass depend on each other
 * (q/k/v -> o_proj -> gate/up -> down_proj -> the next layer's q/k/v), so every launch boundary is a
 * full stop: ~3 us of dispatch ramp, argument reads, codebook staging, first-byte latency and drain
 * per group against 2-10 us of streaming (the reference pays 1-3 such boundaries PER OP,
 * squeezellm/quant_cuda_kernel.cu:157-179, :510-577).  A pass keeps the SAME semantics -- group g + 1
 * sees `vec` (and `mul`) exactly as groups 0..g left them -- but only `vec` waits: long-lived
 * workgroups walk the pass's work items in order, read a group's descriptors, codebooks and first
 * weights while the previous group is still running, and stop at a per-group arrival counter right
 * where `vec` is first consumed.  Weights, lookup tables and the sparse structure (rows / cols / vals /
 * full_rows / full_row_indices) are constants of the pass: they are read AHEAD of the dependency and
 * must not be written by it.
 *
 * Use:   bytes = sqllm_pass_workspace_bytes(ops, group_sizes, n_groups);     (device memory, caller-owned)
 *        sqllm_pass_build(ops, group_sizes, n_groups, workspace, bytes, &pass);   once; blocks (one copy)
 *        sqllm_pass_launch(&pass, stream);                                    per token; graph-capturable
 * A launch is one tiny zeroing kernel (status words, ticket, arrival counters) + the pass kernel.  `pass` is plain data describing
 * the workspace: the library retains nothing.  One launch of a given workspace at a time.
 * Requirements: every op batch <= 1 (matvec ops), all ops of the pass the same bit width, the members of
 * a group share vec and K (as for sqllm_launch_group, up to SQLLM_PASS_MAX_GROUP_OPS per group).
 * Every wait inside the kernel is bounded (option "pass_timeout_ms"): a launch that gives up raises
 * a status word that sqllm_pass_status reads back.
 * Options (sqllm_set_option of the measurement library): "pass_poll_sleep" s_sleep(2) units between two polls of a
 * gate (default 4), "pass_timeout_ms" (default 2000), "pass_wgs_per_cu" workgroups per CU the kernel is launched
 * with (default 0 = the occupancy query's answer).
 * ------------------------------------------------------------------------------------------- */
#define SQLLM_PASS_MAX_GROUP_OPS 8
typedef struct sqllm_pass {
  void* workspace;          /* device */
  int64_t workspace_bytes;
  int64_t segs_offset;      /* byte offsets of the tables inside the workspace */
  int64_t items_offset;
  int32_t state_bytes;      /* leading bytes zeroed before every launch (status words + arrival counters) */
  int32_t bits;
  int32_t n_groups, n_ops, n_items;
  int32_t grid;             /* resident workgroups the kernel is launched with */
  int32_t poll_sleep;       /* option "pass_poll_sleep" at build time */
  int32_t timeout_ms;       /* option "pass_timeout_ms" at build time */
} sqllm_pass;

int64_t sqllm_pass_workspace_bytes(const sqllm_op* ops, const int32_t* group_sizes, int32_t n_groups);
/* Plan only (no device access; GPU-less tests): writes the image the workspace will hold to
 * `host_image` (workspace_bytes bytes, host memory) for a workspace at device address `workspace`. */
int sqllm_pass_plan(const sqllm_op* ops, const int32_t* group_sizes, int32_t n_groups, void* workspace,
                    int64_t workspace_bytes, void* host_image, sqllm_pass* pass);
/* Plan + copy the image into `workspace` (hipMemcpy: blocks; not inside a stream capture). */
int sqllm_pass_build(const sqllm_op* ops, const int32_t* group_sizes, int32_t n_groups, void* workspace,
                     int64_t workspace_bytes, sqllm_pass* pass);
int sqllm_pass_launch(const sqllm_pass* pass, sqllm_stream_t stream);
/* Blocks: synchronises `stream`, then reads the status words of the last launch.  *error: 0 = every gate
 * opened, 1 = a gate timed out (*item = the work item that gave up first; results are not valid). */
int sqllm_pass_status(const sqllm_pass* pass, sqllm_stream_t stream, int32_t* error, int32_t* item);
/* Measurement aid (cf. sqllm_profile_groups): `reps` launches, each bracketed by its own event pair;
 * *avg_us = the kernel's average device-side duration.  Blocks the host. */
int sqllm_pass_profile(const sqllm_pass* pass, sqllm_stream_t stream, int32_t reps, float* avg_us);

#ifdef __cplusplus
}
#endif
#endif /* SQLLM_PASS_API_H */
