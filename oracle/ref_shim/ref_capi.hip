// ORACLE TEST INFRASTRUCTURE -- not product code, never linked into squeezellm_amd.
//
// C-ABI doorway onto the reference's own launchers.  The reference translation unit
// (/root/reference/squeezellm/quant_cuda_kernel.cu) is #included verbatim from where it lies --
// nothing is copied into this repository -- and compiled for gfx950 with the stand-in headers in
// this directory (oracle/build_ref.sh).  The resulting oracle/_ref/libsqllm_ref.so lets the tests
// run the reference kernels on the MI355X next to ours and lets
// tests/golden/make_refkernel_golden.py record their outputs.
//
// Each refk_* function builds (pointer, shape) views and calls the reference host launcher of the
// same name (quant_cuda_kernel.cu:132-738).  The reference launches on the legacy default stream
// with no error checks; we synchronise and return hipGetLastError() so a failing launch is seen.
#include "squeezellm/quant_cuda_kernel.cu"

using torch::Tensor;

// refk_set_sync(0): leave the launches asynchronous, as the reference does (for timing a whole pass;
// tests/ref_kernel_bench.py); the default synchronises after every call.
static int g_sync = 1;
extern "C" void refk_set_sync(int on) { g_sync = on; }

static int finish() {
  if (!g_sync) return (int)hipGetLastError();
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return (int)e;
  return (int)hipGetLastError();
}

extern "C" {

// dense: (vec, mat, mul, lookup_table).  rows_q = qweight rows, N = columns, K = infeatures.
int refk_dense(int bits, int batch, const float* vec, const int* mat, float* mul, const float* lut,
               int K, int N) {
  int rows_q = K / 32 * bits;
  Tensor tmat(mat, rows_q, N), tlut(lut, N, 1 << bits);
  if (batch <= 0) {
    Tensor tvec(vec, K), tmul(mul, N);
    if (bits == 3) vecquant3matmul_nuq_perchannel_cuda(tvec, tmat, tmul, tlut);
    else vecquant4matmul_nuq_perchannel_cuda(tvec, tmat, tmul, tlut);
  } else {
    Tensor tvec(vec, batch, K), tmul(mul, batch, N);
    if (bits == 3) vecquant3matmul_nuq_perchannel_batched_cuda(tvec, tmat, tmul, tlut);
    else vecquant4matmul_nuq_perchannel_batched_cuda(tvec, tmat, tmul, tlut);
  }
  return finish();
}

// spmv: (rows, cols, vals, vec, mul, num_rows, qweight, lookup_table)
int refk_spmv(int bits, int batch, const int* rows, const int* cols, const float* vals, int nnz,
              const float* vec, float* mul, int num_rows, const int* mat, const float* lut, int K,
              int N) {
  int rows_q = K / 32 * bits;
  Tensor trows(rows, num_rows + 1), tcols(cols, nnz), tvals(vals, nnz);
  Tensor tmat(mat, rows_q, N), tlut(lut, N, 1 << bits);
  if (batch <= 0) {
    Tensor tvec(vec, K), tmul(mul, N);
    if (bits == 3) vecquant3matmul_spmv_nuq_perchannel_cuda(trows, tcols, tvals, tvec, tmul, num_rows, tmat, tlut);
    else vecquant4matmul_spmv_nuq_perchannel_cuda(trows, tcols, tvals, tvec, tmul, num_rows, tmat, tlut);
  } else {
    Tensor tvec(vec, batch, K), tmul(mul, batch, N);
    if (bits == 3) vecquant3matmul_spmv_nuq_perchannel_batched_cuda(trows, tcols, tvals, tvec, tmul, num_rows, tmat, tlut);
    else vecquant4matmul_spmv_nuq_perchannel_batched_cuda(trows, tcols, tvals, tvec, tmul, num_rows, tmat, tlut);
  }
  return finish();
}

// hybrid: (rows, cols, vals, vec, full_rows, full_row_indices, mul, num_rows, qweight, lookup_table)
int refk_hybrid(int bits, int batch, const int* rows, const int* cols, const float* vals, int nnz,
                const float* vec, const float* full_rows, const int* full_row_indices, int topX,
                float* mul, int num_rows, const int* mat, const float* lut, int K, int N) {
  int rows_q = K / 32 * bits;
  Tensor trows(rows, num_rows + 1), tcols(cols, nnz), tvals(vals, nnz);
  Tensor tfr(full_rows, K, topX), tfi(full_row_indices, topX);
  Tensor tmat(mat, rows_q, N), tlut(lut, N, 1 << bits);
  if (batch <= 0) {
    Tensor tvec(vec, K), tmul(mul, N);
    if (bits == 3) vecquant3matmul_spmv_hybrid_nuq_perchannel_cuda(trows, tcols, tvals, tvec, tfr, tfi, tmul, num_rows, tmat, tlut);
    else vecquant4matmul_spmv_hybrid_nuq_perchannel_cuda(trows, tcols, tvals, tvec, tfr, tfi, tmul, num_rows, tmat, tlut);
  } else {
    Tensor tvec(vec, batch, K), tmul(mul, batch, N);
    if (bits == 3) vecquant3matmul_spmv_hybrid_nuq_perchannel_batched_cuda(trows, tcols, tvals, tvec, tfr, tfi, tmul, num_rows, tmat, tlut);
    else vecquant4matmul_spmv_hybrid_nuq_perchannel_batched_cuda(trows, tcols, tvals, tvec, tfr, tfi, tmul, num_rows, tmat, tlut);
  }
  return finish();
}

}  // extern "C"
