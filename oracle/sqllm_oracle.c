/* CPU ORACLE (test infrastructure, NOT product code): plain-C restatement of SqueezeLLM's
 * dense-and-sparse LUT-quantised matvec, used (a) by tests as an independent checker next to
 * oracle/sqllm_oracle.py and (b) by bench.py's `cpu_baseline` leg as the timed CPU port.
 * Nothing under squeezellm_amd/ may link or call this.
 *
 * Reference lines followed (paths relative to /root/reference):
 *   4-bit unpack      squeezellm/quant_cuda_kernel.cu:863-877
 *   3-bit unpack      squeezellm/quant_cuda_kernel.cu:776-825 (straddlers :792-793, :809-810)
 *   LUT indexing      squeezellm/quant_cuda_kernel.cu:759-762, :849-852  lookup_table[col*2^b + val]
 *   batched indexing  squeezellm/quant_cuda_kernel.cu:923/:977, :1017/:1036
 *   CSR SpMV          squeezellm/quant_cuda_kernel.cu:1049-1058, batched :1072-1088
 *   top-X rows        squeezellm/quant_cuda_kernel.cu:1101-1121, batched :1139-1162
 *   accumulate-into-mul semantics  squeezellm/quant.py:214-219, :316-318
 *
 * Parity pinning: see the header of oracle/sqllm_oracle.py (pack2 golden vectors + outputs of the
 * reference kernels recorded on an MI355X).
 *
 * Build: make -C oracle   ->  oracle/libsqllm_oracle.so   (gcc -O3 -fopenmp)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* index of weight (k, n), decoded exactly like the kernels do */
static inline unsigned idx4(const uint32_t* q, int N, int k, int n) {
  uint32_t tmp = q[(size_t)(k >> 3) * N + n];
  return (tmp >> (4 * (k & 7))) & 0xf; /* :866-873 */
}

static inline void unpack3_group(uint32_t tmp1, uint32_t tmp2, uint32_t tmp3, uint8_t out[32]) {
  /* one 3-row / 32-weight group, statement for statement after :777-824 */
  uint32_t tmp;
  for (int j = 0; j < 10; ++j) out[j] = (tmp1 >> (3 * j)) & 0x7;
  tmp = (tmp1 >> 30) | ((tmp2 << 2) & 0x4); /* :792 */
  tmp2 >>= 1;                               /* :793 */
  out[10] = tmp & 0x7;
  for (int j = 0; j < 10; ++j) out[11 + j] = (tmp2 >> (3 * j)) & 0x7;
  tmp = (tmp2 >> 30) | ((tmp3 << 1) & 0x6); /* :809 */
  tmp3 >>= 2;                               /* :810 */
  out[21] = tmp & 0x7;
  for (int j = 0; j < 10; ++j) out[22 + j] = (tmp3 >> (3 * j)) & 0x7;
}

/* qweight [K/32*bits, N] -> idx [K, N] (uint8) */
int sqo_unpack(int bits, const int32_t* qweight, int K, int N, uint8_t* idx) {
  const uint32_t* q = (const uint32_t*)qweight;
  if (bits == 4) {
    if (K % 8) return -1;
    for (int k = 0; k < K; ++k)
      for (int n = 0; n < N; ++n) idx[(size_t)k * N + n] = (uint8_t)idx4(q, N, k, n);
    return 0;
  }
  if (bits == 3) {
    if (K % 32) return -1;
    uint8_t g[32];
    for (int grp = 0; grp < K / 32; ++grp)
      for (int n = 0; n < N; ++n) {
        unpack3_group(q[(size_t)(3 * grp) * N + n], q[(size_t)(3 * grp + 1) * N + n],
                      q[(size_t)(3 * grp + 2) * N + n], g);
        for (int j = 0; j < 32; ++j) idx[(size_t)(grp * 32 + j) * N + n] = g[j];
      }
    return 0;
  }
  return -2;
}

/* out[b, n] = mul[b, n] + dense + csr + top-X, accumulated in double.
 * batch <= 0 means the un-batched op (vec [K], mul [N]); it is computed as batch = 1.
 * rows == NULL -> no CSR term; full_rows == NULL -> no top-X term.
 * Returns 0, or a negative code for unsupported shapes. */
int sqo_matvec(int bits, int batch, const float* vec, const int32_t* qweight, const float* mul,
               const float* lut, int K, int N, const int32_t* rows, const int32_t* cols,
               const float* vals, const float* full_rows, const int32_t* full_row_indices, int topX,
               double* out) {
  const uint32_t* q = (const uint32_t*)qweight;
  const int B = batch <= 0 ? 1 : batch;
  const int L = 1 << bits;
  if (bits != 3 && bits != 4) return -2;
  if (K % 32) return -1;
  for (size_t i = 0; i < (size_t)B * N; ++i) out[i] = (double)mul[i];

  /* dense term: one output column at a time so the column's LUT stays in registers/L1 */
#pragma omp parallel for schedule(static)
  for (int n = 0; n < N; ++n) {
    const float* l = lut + (size_t)n * L;
    double* acc = (double*)calloc((size_t)B, sizeof(double));
    if (bits == 4) {
      for (int r = 0; r < K / 8; ++r) {
        uint32_t tmp = q[(size_t)r * N + n];
        for (int j = 0; j < 8; ++j) {
          double w = (double)l[(tmp >> (4 * j)) & 0xf];
          for (int b = 0; b < B; ++b) acc[b] += w * (double)vec[(size_t)b * K + 8 * r + j];
        }
      }
    } else {
      uint8_t g[32];
      for (int grp = 0; grp < K / 32; ++grp) {
        unpack3_group(q[(size_t)(3 * grp) * N + n], q[(size_t)(3 * grp + 1) * N + n],
                      q[(size_t)(3 * grp + 2) * N + n], g);
        for (int j = 0; j < 32; ++j) {
          double w = (double)l[g[j]];
          for (int b = 0; b < B; ++b) acc[b] += w * (double)vec[(size_t)b * K + 32 * grp + j];
        }
      }
    }
    for (int b = 0; b < B; ++b) out[(size_t)b * N + n] += acc[b];
    free(acc);
  }

  /* CSR term (:1049-1058): row r of the CSR is output channel r */
  if (rows) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int r = 0; r < N; ++r) {
      for (int b = 0; b < B; ++b) {
        double dot = 0;
        for (int i = rows[r]; i < rows[r + 1]; ++i) dot += (double)vals[i] * (double)vec[(size_t)b * K + cols[i]];
        out[(size_t)b * N + r] += dot;
      }
    }
  }

  /* top-X term (:1101-1121): serial over c because full_row_indices may repeat */
  if (full_rows) {
    for (int c = 0; c < topX; ++c) {
      int dst = full_row_indices[c];
      for (int b = 0; b < B; ++b) {
        double res = 0;
        for (int k = 0; k < K; ++k) res += (double)full_rows[(size_t)k * topX + c] * (double)vec[(size_t)b * K + k];
        out[(size_t)b * N + dst] += res;
      }
    }
  }
  return 0;
}

int sqo_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
