// Standalone timing harness for the wide matrix-core kernel and text-patched ablations of it (tools/experiments/wide_ablate/run.sh
// builds one binary per variant from a patched COPY of csrc/sqllm_mfma_wide.hip: the product source carries no switches).
// 13B gate/up shape, 4-bit or 3-bit, 2048 rows, random operands; prints microseconds per launch.
#include KERNEL_SOURCE
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  const int bits = argc > 1 ? atoi(argv[1]) : 4;
  const int batch = argc > 2 ? atoi(argv[2]) : 2048;
  const int has_lo = argc > 3 ? atoi(argv[3]) : 0;
  const int k_slices = argc > 4 ? atoi(argv[4]) : 2;
  const int K = 5120, N = 13824;
  const int L = 1 << bits;
  const size_t qwords = (size_t)K / 32 * bits * N;
  std::vector<uint32_t> hq(qwords);
  uint32_t st = 12345;
  for (auto& w : hq) { st = st * 1664525u + 1013904223u; w = st; }
  std::vector<float> hl((size_t)N * L);
  for (auto& v : hl) { st = st * 1664525u + 1013904223u; v = ((int)(st >> 8) % 2001 - 1000) * 2e-5f; }
  const size_t chunks = sqllm::split_planes_chunks(batch, K);
  std::vector<uint16_t> hp(8 * chunks);
  for (size_t i = 0; i < hp.size(); ++i) { st = st * 1664525u + 1013904223u; hp[i] = (i / 8 / 192) % (K / 32 + 1) == (size_t)(K / 32) ? 0 : (uint16_t)(0x3c00 + (st >> 24)); }
  std::vector<uint32_t> hf(sqllm::kSplitFlagWgs, has_lo ? 1u : 0u);
  uint32_t* dq; float* dl; uint16_t* dp; uint32_t* df; float* dy;
  CK(hipMalloc(&dq, qwords * 4)); CK(hipMalloc(&dl, hl.size() * 4)); CK(hipMalloc(&dp, hp.size() * 2)); CK(hipMalloc(&df, hf.size() * 4));
  CK(hipMalloc(&dy, (size_t)batch * N * 4));
  CK(hipMemcpy(dq, hq.data(), qwords * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dl, hl.data(), hl.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dp, hp.data(), hp.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(df, hf.data(), hf.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dy, 0, (size_t)batch * N * 4));
  sqllm::GroupArgs ga;
  memset(&ga, 0, sizeof(ga));
  ga.n_seg = 1;
  sqllm::Segment& sg = ga.seg[0];
  sg.q = dq; sg.y = dy; sg.lut = dl;
  sqllm::KernelGeom& gm = sg.gm;
  gm.K = K; gm.N = N; gm.batch = batch;
  gm.col_tiles = (N + 63) / 64;
  gm.units_total = bits == 4 ? K / 8 : K / 32;
  // k_slices > 0: every unit in that many slices (the old 2-D plan); 0: the product's plan -- whole rounds over all of K, the last round sliced
  const int col_groups = (gm.col_tiles + WT - 1) / WT, row_blocks = (batch + 63) / 64;
  const int slots = 256 * (8 / WT);  // workgroups resident at once
  const int units = col_groups * row_blocks;
  int full = 0;
  if (k_slices > 0) {
    gm.units_per_wg = ((gm.units_total + k_slices - 1) / k_slices + 3) / 4 * 4;
  } else {
    full = units / slots * slots;
    const int rem = units - full;
    const int sl = rem ? slots / rem : 1;
    gm.units_per_wg = ((gm.units_total + sl - 1) / sl + 3) / 4 * 4;
  }
  gm.k_slices = (gm.units_total + gm.units_per_wg - 1) / gm.units_per_wg;
  gm.dense_blocks = full + (units - full) * gm.k_slices;
  dim3 grid(gm.dense_blocks);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0, 0));
    if (bits == 4) hipLaunchKernelGGL((sqllm::sqllm_fused_wide<4, true>), grid, dim3(WT * 64), 0, 0, (const void*)dp, (const uint32_t*)df, full, (float*)nullptr, ga);
    else hipLaunchKernelGGL((sqllm::sqllm_fused_wide<3, true>), grid, dim3(WT * 64), 0, 0, (const void*)dp, (const uint32_t*)df, full, (float*)nullptr, ga);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep && ms < best) best = ms;
  }
  const double products = has_lo ? 6 : 5;
  printf("%-28s bits %d rows %d has_lo %d k_slices %d full %d grid %u : %8.1f us  (%.0f dense TFLOP/s, matrix pipe %.0f TFLOP/s)\n", VARIANT, bits, batch, has_lo,
         gm.k_slices, full, grid.x, best * 1e3, 2.0 * batch * K * N / best / 1e9, 2.0 * products * batch * K * N / best / 1e9);
  return 0;
}
