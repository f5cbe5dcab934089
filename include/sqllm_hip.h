/* sqllm_hip.h -- C ABI of libsqllm_hip.so: MI355X (gfx950) implementation of SqueezeLLM's
 * dense-and-sparse LUT-quantised matvec operator family.
 *
 * This header is the drop-in boundary.  It replaces the reference's pybind11 extension
 * `quant_cuda` (/root/reference/squeezellm/quant_cuda.cpp:112-270, launchers in
 * squeezellm/quant_cuda_kernel.cu:132-738): every `sqllm_vecquant*` entry point below has the
 * name, argument order and meaning of the reference function it replaces, with each
 * torch::Tensor argument replaced by a device pointer and the sizes the reference reads from it
 * via `.size()` passed explicitly after the tensor arguments, plus the HIP stream to launch on.
 *
 * Contract common to all entry points (reference behaviour: SURVEY.md section 8(b)):
 *   - all pointers are DEVICE pointers on the current HIP device; tensors are contiguous,
 *     row-major; `vec`, `mul`, `lookup_table`, `vals`, `full_rows` are fp32; `mat*` (qweight),
 *     `rows`, `cols`, `full_row_indices` are int32;
 *   - `mul` is ACCUMULATED INTO, never overwritten (the caller pre-loads bias or zeros:
 *     squeezellm/quant.py:214-219, :316-318);
 *   - qweight is int32 [height, width] = [K/32*bits, N]; lookup_table is [N, 2^bits];
 *   - the callee retains nothing and never synchronises; it enqueues its kernels on `stream` (NULL = the
 *     legacy default stream, which is what the reference used; the Python binding passes torch's current
 *     stream so calls are graph-capturable): exactly ONE kernel and no allocation for batch 1 and the batch
 *     tiles (up to mfma_min_batch - 1 rows, and every dense-only op up to 16 rows); the `_ws` entry points
 *     (sqllm_launch_ws ...) with a workspace of sqllm_workspace_bytes allocate nothing at ANY batch -- what a
 *     batch needs beside its operands comes out of the caller's workspace, as the reference's launchers
 *     allocate nothing (quant_cuda_kernel.cu:580-657) -- and with a NULL workspace nothing up to 16 rows;
 *     the workspace-less names (sqllm_launch, sqllm_launch_group(s), the reference operator names) take
 *     stream-ordered scratch instead wherever a workspace would have been used (see sqllm_launch);
 *   - return value: 0 on success, a negative SQLLM_E_* code for rejected arguments (the reference
 *     validated nothing and read out of bounds instead), or a positive hipError_t from the launch.
 *
 * Shape requirements: K % 32 == 0 and N % 4 == 0 (the reference needed K % 128 == 0 and
 * N % 128 == 0: quant_cuda_kernel.cu:754,776,841-852).  qweight must be 16-byte aligned.
 */
#ifndef SQLLM_HIP_H
#define SQLLM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SQLLM_ABI_VERSION 1

/* error codes (negative; positive return values are hipError_t) */
#define SQLLM_OK 0
#define SQLLM_E_BITS (-1)      /* bits not in {3, 4} (quant.py:42-43) */
#define SQLLM_E_SHAPE (-2)     /* K % 32 != 0, N % 4 != 0, height != K/32*bits, non-positive dims */
#define SQLLM_E_NULL (-3)      /* a required pointer is NULL */
#define SQLLM_E_ALIGN (-4)     /* qweight not 16-byte aligned */
#define SQLLM_E_SPARSE (-5)    /* inconsistent sparse operands (nnz < 0, num_rows != N, topX < 0; with "validate_csr": bad rows[]) */
#define SQLLM_E_BATCH (-6)     /* batch < 1 or vec_height != K for a batched op */
#define SQLLM_E_OPTION (-7)    /* unknown option name / bad value */
#define SQLLM_E_GROUP (-8)     /* group of 0 or > 4 ops, or members differ in vec / K / bits / batch */

typedef void* sqllm_stream_t; /* a hipStream_t */

/* ---------------------------------------------------------------------------------------------
 * Generic descriptor form.  All twelve named entry points are thin adapters over sqllm_launch.
 * ------------------------------------------------------------------------------------------- */
typedef struct sqllm_op {
  int32_t bits;  /* 3 or 4 */
  int32_t batch; /* 0: matvec op, vec [K], mul [N].  >= 1: *_batched op, vec [batch, K], mul [batch, N] */
  int32_t K;     /* infeatures  */
  int32_t N;     /* outfeatures */
  const float* vec;
  const int32_t* qweight;     /* [K/32*bits, N] */
  float* mul;                 /* accumulated into */
  const float* lookup_table;  /* [N, 2^bits] */
  /* CSR outliers; rows == NULL -> no sparse term */
  const int32_t* rows; /* [N + 1] */
  const int32_t* cols; /* [nnz]   */
  const float* vals;   /* [nnz]   */
  int32_t nnz;
  /* "top-X" dense rows; full_rows == NULL -> no such term */
  int32_t topX;
  const float* full_rows;           /* [K, topX] */
  const int32_t* full_row_indices;  /* [topX]    */
} sqllm_op;

/* Enqueue  mul += W_lut . vec (+ CSR . vec) (+ full_rows^T . vec scattered)  on `stream`: one fused
 * kernel up to 16 rows -- from mfma_min_batch rows on, an op WITH sparse terms gets a small kernel in
 * front of it that writes vec transposed (and split into bf16 planes) into stream-ordered scratch
 * (hipMallocAsync / hipFreeAsync on `stream`; never inside a stream capture: the sparse terms gather
 * from vec itself there).  A wider batch is up to four kernels -- a transpose of vec into such scratch
 * (only with a CSR term), the sparse terms, (wide form only: "mfma_wide_min_batch") the split of vec
 * into bf16 planes, again in such scratch, and the dense term on the matrix cores; inside a stream
 * capture that scratch becomes memory nodes of the graph unless option "scratch_in_capture" is 0.
 * No host synchronisation in any case.  On first use of the scratch per device the release threshold of
 * the device's default memory pool is raised (option "scratch_pool_threshold").
 * Callers that own a workspace use sqllm_launch_ws instead: nothing is allocated then. */
int sqllm_launch(const sqllm_op* op, sqllm_stream_t stream);

/* The same with a CALLER-OWNED workspace -- the form that keeps the reference's contract at every batch
 * (its launchers allocate nothing: quant_cuda_kernel.cu:580-657): `workspace` is `workspace_bytes` of
 * device memory, 16-byte aligned, contents irrelevant before and after; sqllm_workspace_bytes(ops, n)
 * says how much the op (n = 1) or the group can use (0: none -- batch 1, batch tiles).  It serves one
 * launch at a time: launches on one stream may share it, concurrent streams may not.  What lives in it:
 *   mfma_min_batch..16 rows (fused small launch: 4-bit from 7 rows, 3-bit from 9; only with sparse terms)  vec transposed, xT[k][rows rounded up to 8 / 16]: the
 *            CSR walk of the dense workgroups and the top-X slabs then read one cache line per k for all batch rows
 *            instead of one per row; behind it vec as three bf16 planes in fragment order ((K / 32 + 1) x 3 KB): the
 *            dense term loads its operands already split (one small kernel in front of the launch writes both);
 *   17+ rows  what sqllm_launch takes from stream-ordered scratch: vec transposed, its bf16 planes, the
 *            wide form's slabs.
 * A NULL or too small workspace is not an error: 2..16 rows then gather from vec itself (ONE kernel, no
 * allocation, the default memory pool untouched), 17+ rows fall back to the stream-ordered scratch of
 * sqllm_launch.  A stream capture of a `_ws` launch with a sufficient workspace contains kernel nodes only. */
int64_t sqllm_workspace_bytes(const sqllm_op* ops, int32_t n_ops);
int sqllm_launch_ws(const sqllm_op* op, void* workspace, int64_t workspace_bytes, sqllm_stream_t stream);

/* Enqueue `n_ops` ops back to back on `stream` from one host call (a decode pass over a stack of
 * QuantLinearLUT layers costs one FFI crossing instead of n_ops).  Stops at the first error and
 * returns it; *n_done (may be NULL) receives the number of ops enqueued. */
int sqllm_launch_sequence(const sqllm_op* ops, int32_t n_ops, sqllm_stream_t stream, int32_t* n_done);

/* Same-input fusion: ONE kernel for 1..4 ops that read the same `vec` (same K, bits and batch) --
 * in a decoder layer q_proj/k_proj/v_proj, and gate_proj/up_proj (squeezellm/model_parse.py:53-61).
 * Every op keeps its own qweight / lookup_table / mul / sparse operands; the launch's workgroups
 * are simply divided between them.  Halves the launch count of a LLaMA decode pass, which matters
 * because each launch carries ~2-3 us of fixed cost against 1-4 us of streaming. */
int sqllm_launch_group(const sqllm_op* ops, int32_t n_ops, sqllm_stream_t stream);
int sqllm_launch_group_ws(const sqllm_op* ops, int32_t n_ops, void* workspace, int64_t workspace_bytes, sqllm_stream_t stream);

/* A whole pass as consecutive groups: group g covers the next group_sizes[g] entries of `ops`.
 * *n_done (may be NULL) receives the number of groups enqueued. */
int sqllm_launch_groups(const sqllm_op* ops, const int32_t* group_sizes, int32_t n_groups,
                        sqllm_stream_t stream, int32_t* n_done);
/* ... with one workspace for all of them (they run one after the other on `stream`): the largest
 * sqllm_workspace_bytes of any group serves the pass */
int sqllm_launch_groups_ws(const sqllm_op* ops, const int32_t* group_sizes, int32_t n_groups, void* workspace,
                           int64_t workspace_bytes, sqllm_stream_t stream, int32_t* n_done);

/* Measurement aid (used by bench.py's roofline leg, never on the serving path): enqueue the ops like
 * sqllm_launch_sequence, but attach a start/stop event pair to EVERY kernel dispatch
 * (hipExtLaunchKernelGGL), so that each kernel's own device-side duration -- the quantity
 * rocprofv3 --kernel-trace reports -- is available without a profiler.  Runs `reps` passes,
 * synchronises `stream` after each, and writes the per-op average in microseconds to
 * avg_us[0..n_ops).  Blocks the host; not graph-capturable. */
int sqllm_profile_sequence(const sqllm_op* ops, int32_t n_ops, sqllm_stream_t stream, int32_t reps,
                           float* avg_us);
/* the same for grouped launches: avg_us[0..n_groups) */
int sqllm_profile_groups(const sqllm_op* ops, const int32_t* group_sizes, int32_t n_groups,
                         sqllm_stream_t stream, int32_t reps, float* avg_us);
int sqllm_profile_groups_ws(const sqllm_op* ops, const int32_t* group_sizes, int32_t n_groups, void* workspace,
                            int64_t workspace_bytes, sqllm_stream_t stream, int32_t reps, float* avg_us);

/* ---------------------------------------------------------------------------------------------
 * Fused linear: the whole matvec branch of QuantLinearLUT.forward in one kernel.
 *
 * The reference wraps every operator call in three more launches -- `y = bias.clone()` or
 * `torch.zeros`, `x.float()`, `y.to(fp16)` (squeezellm/quant.py:214-223, :311-312; batched:
 * :314-321, :380-383).  sqllm_linear_f16 takes the activations as fp16 and writes fp16:
 *
 *     out[b, n] = fp16( bias[n] + sum_k W[n, k] * float(x[b, k]) )      (all three weight terms)
 *
 * `op` is read as for sqllm_launch except that  op.vec  is  const _Float16* [batch, K]  and
 * op.mul  is  _Float16* [batch, N], OVERWRITTEN (not accumulated into).  Accumulation is fp32.
 * `workspace`: sqllm_linear_workspace_bytes(&op) bytes of device memory, 16-byte aligned, that the
 * caller zero-fills ONCE; every launch leaves it zero-filled again.  A workspace serves one
 * launch at a time (launches on one stream may share it; concurrent streams may not).
 * Like every entry point: one kernel, nothing allocated, nothing retained, no synchronisation.
 * Accumulation runs in 2^-28 fixed point inside the workspace (so that one returning atomic both
 * deposits a partial sum and counts it): every partial sum is clamped to +-131072 (2 x the
 * largest finite fp16), i.e. results are exact to well below one fp16 ulp wherever the fp16 result
 * is finite.  Non-finite partial sums are carried by three sticky flag bits of the accumulator word:
 * a column that received a NaN comes out NaN, one that received +inf / -inf comes out +inf / -inf
 * (NaN if infinities of both signs met) -- the same pattern the operator path's fp32 atomics produce.  The CSR operands must be consistent (rows[N] == nnz, rows non-decreasing):
 * completion is detected by counting the contributions `rows` announces.  Shapes whose columns
 * could receive more than 63 partial sums (K slices + K / 1024 + 2 CSR chunks) are rejected with
 * SQLLM_E_SHAPE.
 * ------------------------------------------------------------------------------------------- */
typedef struct sqllm_linear {
  sqllm_op op;
  const float* bias; /* fp32 [N] or NULL */
  void* workspace;
} sqllm_linear;

int64_t sqllm_linear_workspace_bytes(const sqllm_op* op);
int sqllm_linear_f16(const sqllm_linear* lin, sqllm_stream_t stream);
/* a pass of fused linears as consecutive same-input groups (cf. sqllm_launch_groups); the members
 * of a group share op.vec, K, bits and batch and each has its own workspace */
int sqllm_linear_f16_groups(const sqllm_linear* lins, const int32_t* group_sizes, int32_t n_groups,
                            sqllm_stream_t stream, int32_t* n_done);

/* ---------------------------------------------------------------------------------------------
 * The reference operator names.
 * height/width = mat.size(0)/mat.size(1) of the qweight tensor (quant_cuda_kernel.cu:138-139).
 * ------------------------------------------------------------------------------------------- */

/* replaces vecquant3matmul_nuq_perchannel / vecquant4matmul_nuq_perchannel
 * (quant_cuda.cpp:112-125 -> quant_cuda_kernel.cu:132-154 / :157-179) */
int sqllm_vecquant3matmul_nuq_perchannel(const float* vec, const int32_t* mat, float* mul,
                                         const float* lookup_table, int height, int width,
                                         sqllm_stream_t stream);
int sqllm_vecquant4matmul_nuq_perchannel(const float* vec, const int32_t* mat, float* mul,
                                         const float* lookup_table, int height, int width,
                                         sqllm_stream_t stream);

/* replaces vecquant{3,4}matmul_nuq_perchannel_batched (quant_cuda.cpp:126-139 ->
 * quant_cuda_kernel.cu:182-207 / :210-235); batch = vec.size(0), vec_height = vec.size(1) */
int sqllm_vecquant3matmul_nuq_perchannel_batched(const float* vec, const int32_t* mat, float* mul,
                                                 const float* lookup_table, int height, int width,
                                                 int batch, int vec_height, sqllm_stream_t stream);
int sqllm_vecquant4matmul_nuq_perchannel_batched(const float* vec, const int32_t* mat, float* mul,
                                                 const float* lookup_table, int height, int width,
                                                 int batch, int vec_height, sqllm_stream_t stream);

/* replaces vecquant{3,4}matmul_spmv_nuq_perchannel (quant_cuda.cpp:141-166 ->
 * quant_cuda_kernel.cu:238-281 / :284-327).  `mat` is the CSR value array (the reference's name),
 * `mat3`/`mat4` the packed qweight; nnz = cols.size(0). */
int sqllm_vecquant3matmul_spmv_nuq_perchannel(const int32_t* rows, const int32_t* cols,
                                              const float* mat, const float* vec, float* mul,
                                              int num_rows, const int32_t* mat3,
                                              const float* lookup_table, int height, int width,
                                              int nnz, sqllm_stream_t stream);
int sqllm_vecquant4matmul_spmv_nuq_perchannel(const int32_t* rows, const int32_t* cols,
                                              const float* mat, const float* vec, float* mul,
                                              int num_rows, const int32_t* mat4,
                                              const float* lookup_table, int height, int width,
                                              int nnz, sqllm_stream_t stream);

/* replaces vecquant{3,4}matmul_spmv_nuq_perchannel_batched (quant_cuda.cpp:168-193 ->
 * quant_cuda_kernel.cu:331-382 / :385-435) */
int sqllm_vecquant3matmul_spmv_nuq_perchannel_batched(const int32_t* rows, const int32_t* cols,
                                                      const float* mat, const float* vec,
                                                      float* mul, int num_rows,
                                                      const int32_t* mat3,
                                                      const float* lookup_table, int height,
                                                      int width, int nnz, int batch,
                                                      int vec_height, sqllm_stream_t stream);
int sqllm_vecquant4matmul_spmv_nuq_perchannel_batched(const int32_t* rows, const int32_t* cols,
                                                      const float* mat, const float* vec,
                                                      float* mul, int num_rows,
                                                      const int32_t* mat4,
                                                      const float* lookup_table, int height,
                                                      int width, int nnz, int batch,
                                                      int vec_height, sqllm_stream_t stream);

/* replaces vecquant{3,4}matmul_spmv_hybrid_nuq_perchannel (quant_cuda.cpp:195-224 ->
 * quant_cuda_kernel.cu:439-506 / :510-577); full_rows is [full_height = K, topX] */
int sqllm_vecquant3matmul_spmv_hybrid_nuq_perchannel(
    const int32_t* rows, const int32_t* cols, const float* mat, const float* vec,
    const float* full_rows, const int32_t* full_row_indices, float* mul, int num_rows,
    const int32_t* mat3, const float* lookup_table, int height, int width, int nnz, int topX,
    sqllm_stream_t stream);
int sqllm_vecquant4matmul_spmv_hybrid_nuq_perchannel(
    const int32_t* rows, const int32_t* cols, const float* mat, const float* vec,
    const float* full_rows, const int32_t* full_row_indices, float* mul, int num_rows,
    const int32_t* mat4, const float* lookup_table, int height, int width, int nnz, int topX,
    sqllm_stream_t stream);

/* replaces vecquant{3,4}matmul_spmv_hybrid_nuq_perchannel_batched (quant_cuda.cpp:226-255 ->
 * quant_cuda_kernel.cu:580-657 / :661-738) */
int sqllm_vecquant3matmul_spmv_hybrid_nuq_perchannel_batched(
    const int32_t* rows, const int32_t* cols, const float* mat, const float* vec,
    const float* full_rows, const int32_t* full_row_indices, float* mul, int num_rows,
    const int32_t* mat3, const float* lookup_table, int height, int width, int nnz, int topX,
    int batch, int vec_height, sqllm_stream_t stream);
int sqllm_vecquant4matmul_spmv_hybrid_nuq_perchannel_batched(
    const int32_t* rows, const int32_t* cols, const float* mat, const float* vec,
    const float* full_rows, const int32_t* full_row_indices, float* mul, int num_rows,
    const int32_t* mat4, const float* lookup_table, int height, int width, int nnz, int topX,
    int batch, int vec_height, sqllm_stream_t stream);

/* The two names squeezellm/quant.py:237-250 / :281-294 call for `balanced=True` layers but the
 * reference never defined or exported (quant_cuda.cpp:257-270).  Argument order is quant.py's.
 * `startrows`/`num_threads` describe the reference's intended thread partition; this
 * implementation balances by nnz on its own and ignores them (they may be NULL/0).
 * Semantics = the spmv op: mul += W_lut . vec + CSR . vec. */
int sqllm_vecquant3matmul_spmv_balanced_nuq_perchannel(
    const int32_t* rows, const int32_t* cols, const int32_t* startrows, const float* mat,
    const float* vec, float* mul, const int32_t* mat3, const float* lookup_table, int num_rows,
    int num_threads, int numvals, int height, int width, sqllm_stream_t stream);
int sqllm_vecquant4matmul_spmv_balanced_nuq_perchannel(
    const int32_t* rows, const int32_t* cols, const int32_t* startrows, const float* mat,
    const float* vec, float* mul, const int32_t* mat4, const float* lookup_table, int num_rows,
    int num_threads, int numvals, int height, int width, sqllm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Library services
 * ------------------------------------------------------------------------------------------- */
int sqllm_abi_version(void);
const char* sqllm_error_string(int code); /* static string for SQLLM_E_* and hipError_t values */

/* Launch-geometry knobs (for measurement sweeps; defaults are chosen per shape).  Options are kept
 * PER DEVICE: a set / get applies to the calling thread's current HIP device.
 *   "target_wgs"      dense workgroups to aim for (default 0 = 1 x CU count for layers <= 12 MB,
 *                     3 x CU count above)
 *   "groups_per_wave" force the K units each wave walks (default 0 = derived from target_wgs)
 *   "sparse_last"     1 = CSR / top-X workgroups after the dense ones in the grid (default 0: first)
 *   "cu_count"        override the CU count used for planning (GPU-less tests)
 *   "mfma_min_batch", "cols_min_batch", "cols_max_batch"
 *                     routing of the *_batched operators by batch size: cols_min_batch .. cols_max_batch
 *                     rows run on the column-lane kernel (lane = output column, vec in SGPRs),
 *                     mfma_min_batch rows and more on the matrix cores, everything else on the
 *                     batch tiles of the batch-1 kernel (tiles of exactly 1..8 rows).  Defaults (value 0 =
 *                     measured default, which depends on the bit width; get_option returns the stored 0): 4-bit 2..4 / 7
 *                     (9 for an op of <= 16 MB of packed weights alone in its launch), 3-bit 2..8 / 9.
 *                     At ONE row (batch 1, or the matvec names) the defaults route by launch shape instead (round 6, sqllm_capi.hip: cols_pays_batch1 --
 *                     dense-only launches of >= 16 MB that are a three-op group, a tall single op or a 3-bit two-op group, and the 7B-class sparse groups,
 *                     take the column-lane kernel); an explicit cols_min_batch = 1 sends every one-row launch there, a huge cols_min_batch none.
 *                     With the two cols_* options at their defaults the column-lane kernel is further reserved
 *                     for what it measured faster on: 4-bit, groups of three or more ops (up to 4 rows) and single ops of
 *                     >= 20 MB packed weights (up to 6 rows); 3-bit, >= 16 MB at up to 4 rows or N >= 8192; setting either option
 *                     takes the range at its word.
 *   "cols_groups"     1 (default): a GROUP of ops over one vec (sqllm_launch_group) may take the column-lane kernel
 *                     too, as one launch, judged by the sum of its columns; 0: groups stay on the batch tiles
 *   "sparse_transpose" 1 (default): the sparse terms of a batched op (mfma_min_batch rows and more) read a transposed copy of
 *                     vec (lane = batch row, coalesced) out of the caller's workspace (sqllm_launch_ws) or, for the
 *                     workspace-less names, out of stream-ordered scratch; 0: they gather from vec itself, as they do
 *                     anyway when neither can be had
 *                     The scratch is stream-ordered (hipMallocAsync / hipFreeAsync on the caller's
 *                     stream, K x ceil64(batch) floats per op or group); on first use per device the
 *                     library raises the release threshold of the device's DEFAULT memory pool to
 *                     1 GiB (never lowers it) so that the block survives synchronisations.
 *   "scratch_pool_threshold" 1 (default): on first use of that scratch per device the library raises the release threshold of
 *                     the device's DEFAULT memory pool to 1 GiB (process-wide state, never lowered); 0: it leaves the pool alone
 *                     (every synchronisation may then hand the scratch back to the OS)
 *   "scratch_in_capture" 1 (default): that scratch is also taken while the stream is capturing --
 *                     a captured wide-batch op with a CSR term then carries a memory-allocation
 *                     and a memory-free node in the graph; 0 keeps captures allocation-free (the
 *                     CSR term gathers from vec instead)
 *   "mfma_split"      1 (default): the dense term of a wide batch (mfma_min_batch rows and more) runs on the bf16 matrix
 *                     instructions with every fp32 operand split EXACTLY into three bf16 values and six of the nine partial
 *                     products kept (fp32-class results, 2.7 x the matrix rate of the fp32 instruction); 0: the fp32 matrix
 *                     instruction (bit-for-bit an fp32 FMA chain per output)
 *   "mfma_fuse_small" 1 (default, with mfma_split): from mfma_min_batch up to 16 rows an op -- or a whole GROUP of ops over one vec
 *                     (sqllm_launch_group) -- is ONE launch of the split matrix-core kernel: every dense workgroup walks the CSR
 *                     non-zeros of its own 64 output channels (no chunk workgroups), the top-X slabs ride in the same grid; with a
 *                     workspace (or, eagerly, scratch) a small kernel in front transposes vec for those two (K < 2^26);
 *                     0: one launch per op plus a launch for its sparse terms (as from 17 rows on)
 *   "mfma_fuse_sparse" 1 (default, with mfma_split): from 17 rows up to the wide form an op's CSR / top-X workgroups run in the grid of
 *                     its dense launch -- always up to 64 rows (33-64 rows: as two 32-row passes if they outnumber the CUs),
 *                     beyond that while they are fewer than the CUs; 0: a launch of their own first
 *   "mfma_wide_min_batch"  0 (default): with mfma_split, the WIDE form of that kernel -- workgroups of eight 64-column tiles, one per
 *                     wave, all on the same k's: the vec values of a step are fetched once per workgroup instead of once per tile --
 *                     takes over from 64 rows up once batch * K * N >= 5.7e9 (3-bit: 4e9; three times that while the stream is
 *                     capturing; without scratch: once its units of 64 rows x 8 tiles fill 80 % of the CUs).  13B shapes: 128
 *                     rows 104 -> 85 us, 2048 rows 1.66 -> 0.96 ms.  n > 0: from n rows on, whatever the shape; a huge value:
 *                     never.  Geometry through sqllm_plan_query: grid_y = 1, dense_blocks = workgroups (whole rounds of units
 *                     over all of K + the last round's units in k_slices K slices of groups_per_wave units, whose sums a
 *                     second launch adds to mul).
 *   "split_planes_min_batch"  0 (default: 64): rows from which the wide form takes vec split ONCE into bf16 planes in stream-ordered
 *                     scratch (6 bytes per vec value, rows padded to 64, plus 16 KB per K slice and tile; a split kernel in front of the op; fp16-born vec -- what
 *                     QuantLinearLUT.forward passes -- then costs five partial products instead of six); below, or without
 *                     scratch (scratch_in_capture = 0 while capturing, allocation failure), every wave splits its values in
 *                     registers and K slices add atomically; a huge value: never.
 *   "small_wgs_per_cu" 0 (default: 2): dense workgroups per CU the planner of the fused small launch (mfma_min_batch .. 16 rows)
 *                     aims at -- one round of workgroups, as many as the kernel's registers admit at once
 *   "small_reserve_topx" 0 (default): with a transposed vec at hand the dense ranges of that launch are always planned for the
 *                     slots the (8-24) top-X workgroups leave; 1 does the same without one (one workgroup per top-X slab:
 *                     measured slower, profiles/r05_small_split_reserve.txt)
 *   "small_planes"    1 (default): with a transposed vec at hand the fused small launch also takes vec split into bf16 planes
 *                     (written by the same kernel in front); 0: its dense term splits vec in registers
 *   "validate_csr"    debugging aid, default 0.  1 = before every launch that carries a CSR term,
 *                     check ON THE DEVICE that rows[] is non-decreasing with rows[0] == 0 and
 *                     rows[N] == nnz, and return SQLLM_E_SPARSE otherwise.  Blocks the host (one
 *                     tiny kernel, a 4-byte read-back, a stream synchronise; 4 bytes of stream-ordered
 *                     scratch on the current device per check); skipped while the stream is capturing.
 *                     Meant for the fused linear, which counts on `rows` to detect completion.
 * Returns SQLLM_E_OPTION for an unknown name and for a value outside the option's range: switches take 0 / 1 only,
 * "small_wgs_per_cu" 0..8, "cu_count" up to 65536, "target_wgs" / "groups_per_wave" up to 2^24, the *_min_batch /
 * *_max_batch thresholds any non-negative int.  No value of any option changes a result beyond fp32 round-off. */
int sqllm_set_option(const char* name, int value);
int sqllm_get_option(const char* name, int* value);

/* Geometry the library would use for this op launched ALONE (sqllm_launch / the operator names).  Ops
 * that share a launch (sqllm_launch_group) are planned with other workgroup counts (the launch's
 * workgroup target is divided between them), on the kernel the GROUP routes to: the batch tiles, the
 * column-lane kernel (judged by the sum of the group's columns, option "cols_groups") or, from
 * mfma_min_batch up to 16 rows, the split matrix-core kernel. */
typedef struct sqllm_plan {
  int32_t col_tiles, k_slices, groups_per_wave, dense_blocks, csr_blocks, topx_blocks, grid_x, grid_y;
} sqllm_plan;
int sqllm_plan_query(const sqllm_op* op, sqllm_plan* plan);

#ifdef __cplusplus
}
#endif
#endif /* SQLLM_HIP_H */
