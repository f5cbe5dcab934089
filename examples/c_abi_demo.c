/* c_abi_demo.c -- the drop-in boundary used from plain C, no PyTorch and no Python:
 * the 4-bit dense operator and the fused fp16 linear of include/sqllm_hip.h on seeded operands,
 * checked against a host loop that restates the packed format (qweight row r, bits [4j, 4j+4) =
 * index of k = 8r + j; squeezellm/quant.py:180-184).
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_abi_demo.c \
 *       -Lsqueezellm_amd -lsqllm_hip -L/opt/rocm/lib -lamdhip64 -lm \
 *       -Wl,-rpath,$PWD/squeezellm_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/c_abi_demo && /tmp/c_abi_demo
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sqllm_hip.h"

#define CHECK(e)                                                                     \
  do {                                                                               \
    hipError_t err_ = (e);                                                           \
    if (err_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(err_));    \
      return 2;                                                                      \
    }                                                                                \
  } while (0)

static uint32_t rng_state = 12345u;
static uint32_t rnd(void) { rng_state = rng_state * 1664525u + 1013904223u; return rng_state; }
static float rndf(void) { return (float)(rnd() >> 8) / 16777216.0f - 0.5f; }

/* fp32 -> fp16 bits, round to nearest even (normal range only: the demo's values are small) */
static uint16_t f2h(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  uint32_t sign = (u >> 16) & 0x8000u; int32_t e = (int32_t)((u >> 23) & 0xff) - 127 + 15; uint32_t m = u & 0x7fffffu;
  if (e <= 0) return (uint16_t)sign;
  uint32_t h = (uint32_t)(e << 10) | (m >> 13);
  uint32_t rem = m & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;
  return (uint16_t)(sign | h);
}
static float h2f(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16; int32_t e = (h >> 10) & 0x1f; uint32_t m = h & 0x3ffu;
  if (e == 0) return 0.0f; /* demo values are normal or zero */
  uint32_t u = sign | (uint32_t)((e - 15 + 127) << 23) | (m << 13);
  float f; memcpy(&f, &u, 4); return f;
}

int main(void) {
  const int K = 4096, N = 4096, rows_q = K / 8;
  int32_t* q = (int32_t*)malloc((size_t)rows_q * N * 4);
  float* lut = (float*)malloc((size_t)N * 16 * 4);
  float* x = (float*)malloc((size_t)K * 4);
  float* bias = (float*)malloc((size_t)N * 4);
  double* ref = (double*)calloc((size_t)N, 8);
  float* y = (float*)malloc((size_t)N * 4);
  uint16_t* x16 = (uint16_t*)malloc((size_t)K * 2);
  uint16_t* y16 = (uint16_t*)malloc((size_t)N * 2);
  for (size_t i = 0; i < (size_t)rows_q * N; ++i) q[i] = (int32_t)rnd();
  for (size_t i = 0; i < (size_t)N * 16; ++i) lut[i] = 0.04f * rndf();
  for (int k = 0; k < K; ++k) { x16[k] = f2h(2.0f * rndf()); x[k] = h2f(x16[k]); }  /* same values in both precisions */
  for (int n = 0; n < N; ++n) bias[n] = 0.02f * rndf();
  for (int r = 0; r < rows_q; ++r)
    for (int n = 0; n < N; ++n) {
      uint32_t w = (uint32_t)q[(size_t)r * N + n];
      for (int j = 0; j < 8; ++j) ref[n] += (double)lut[(size_t)n * 16 + ((w >> (4 * j)) & 15u)] * (double)x[8 * r + j];
    }

  int32_t* dq; float *dlut, *dx, *dy, *dbias; uint16_t *dx16, *dy16; void* ws;
  CHECK(hipMalloc((void**)&dq, (size_t)rows_q * N * 4));
  CHECK(hipMalloc((void**)&dlut, (size_t)N * 16 * 4));
  CHECK(hipMalloc((void**)&dx, (size_t)K * 4));
  CHECK(hipMalloc((void**)&dy, (size_t)N * 4));
  CHECK(hipMalloc((void**)&dbias, (size_t)N * 4));
  CHECK(hipMalloc((void**)&dx16, (size_t)K * 2));
  CHECK(hipMalloc((void**)&dy16, (size_t)N * 2));
  CHECK(hipMemcpy(dq, q, (size_t)rows_q * N * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dlut, lut, (size_t)N * 16 * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dx, x, (size_t)K * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dbias, bias, (size_t)N * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dx16, x16, (size_t)K * 2, hipMemcpyHostToDevice));

  /* 1. the reference operator: mul += W . vec, accumulated into a caller-initialised mul (here: bias) */
  CHECK(hipMemcpy(dy, bias, (size_t)N * 4, hipMemcpyHostToDevice));
  int rc = sqllm_vecquant4matmul_nuq_perchannel(dx, dq, dy, dlut, rows_q, N, NULL);
  if (rc) { fprintf(stderr, "operator: %s\n", sqllm_error_string(rc)); return 1; }
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(y, dy, (size_t)N * 4, hipMemcpyDeviceToHost));
  double worst = 0, scale = 0;
  for (int n = 0; n < N; ++n) {
    double e = fabs((double)y[n] - (ref[n] + bias[n]));
    if (e > worst) worst = e;
    if (fabs(ref[n]) > scale) scale = fabs(ref[n]);
  }
  printf("operator     : max |err| / max |y| = %.3g\n", worst / scale);
  if (worst / scale > 2e-5) return 1;

  /* 2. the fused fp16 linear: out = fp16(W . x + bias), one kernel, workspace zero-filled once */
  sqllm_linear lin;
  memset(&lin, 0, sizeof(lin));
  lin.op.bits = 4; lin.op.batch = 0; lin.op.K = K; lin.op.N = N;
  lin.op.vec = (const float*)dx16; lin.op.qweight = dq; lin.op.mul = (float*)dy16; lin.op.lookup_table = dlut;
  lin.bias = dbias;
  int64_t wsb = sqllm_linear_workspace_bytes(&lin.op);
  CHECK(hipMalloc(&ws, (size_t)wsb));
  CHECK(hipMemset(ws, 0, (size_t)wsb));
  lin.workspace = ws;
  for (int rep = 0; rep < 2; ++rep) { /* the second launch runs on the workspace the first one left behind */
    rc = sqllm_linear_f16(&lin, NULL);
    if (rc) { fprintf(stderr, "linear: %s\n", sqllm_error_string(rc)); return 1; }
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(y16, dy16, (size_t)N * 2, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int n = 0; n < N; ++n) {
      double want = ref[n] + bias[n], got = h2f(y16[n]);
      double tol = fmax(fabs(want), 6.2e-5) / 1024.0 + 1e-6; /* one fp16 ulp */
      if (fabs(got - want) > tol) ++bad;
    }
    printf("fused linear : launch %d, outputs off by more than one fp16 ulp: %d of %d\n", rep + 1, bad, N);
    if (bad) return 1;
  }
  printf("ok (ABI version %d)\n", sqllm_abi_version());
  return 0;
}
