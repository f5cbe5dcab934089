// sqllm_ablation.hip -- MEASUREMENT LIBRARY: ablation instantiations of the fused kernel (sqllm_fused.h: ABL bits --
// no lookups, no decode, no staging, no epilogue ...) and the calibration kernels (empty launch, linear and tiled
// streaming reads), selected by the option "ablate".  Installed as the fused launch's variant hook.
#include "sqllm_fused.h"

#ifndef SQLLM_ABLATION_BUILD
#error "csrc/experimental/ belongs to the measurement library (python -m squeezellm_amd.build --ablation)"
#endif

namespace sqllm {

// calibration kernels (measurement builds only): what does this box give an empty launch and a
// plain linear 16-B/lane streaming read of the same bytes?
__global__ void __launch_bounds__(256) sqllm_calib_empty(float* y) {
  if (threadIdx.x == 12345) y[0] = 1.f;
}
template <int UNROLL, bool NT>
__global__ void __launch_bounds__(256) sqllm_calib_stream(const u32x4* q, size_t n16, float* y) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t acc = 0;
  for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
    u32x4 w[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) w[u] = NT ? __builtin_nontemporal_load(q + i + u * stride) : q[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= w[u].x ^ w[u].y ^ w[u].z ^ w[u].w;
  }
  for (; i < n16; i += stride) { u32x4 w = q[i]; acc ^= w.x ^ w.y ^ w.z ^ w.w; }
  if (acc == 0x12345678u) y[0] = 1.f;
}
// tiled streaming read: a wave covers (64 / SEGL) rows x (SEGL lanes x 16 B) per load instruction,
// a workgroup of 4 waves walks `rows_per_wg` rows of one column tile -- how narrow may a row segment
// get before HBM efficiency drops?
template <int SEGL>
__global__ void __launch_bounds__(256) sqllm_calib_tiled(const u32x4* q, int rows_total, int row_stride16,
                                                        int col_tiles, int rows_per_wg, float* y) {
  constexpr int RPI = 64 / SEGL;  // rows per wave-instruction
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ct = blockIdx.x % col_tiles, ks = blockIdx.x / col_tiles;
  int c16 = ct * SEGL + (lane % SEGL);
  if (c16 > row_stride16 - 1) c16 = row_stride16 - 1;
  const int r0 = ks * rows_per_wg;
  int r1 = r0 + rows_per_wg;
  if (r1 > rows_total) r1 = rows_total;
  uint32_t acc = 0;
  // wave w takes rows r0 + w*RPI + lane/SEGL, stepping 4*RPI
  for (int r = r0 + wave * RPI + lane / SEGL; r < r1; r += 4 * RPI * 4) {
    u32x4 w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int rr = r + u * 4 * RPI;
      if (rr > rows_total - 1) rr = rows_total - 1;
      w[u] = __builtin_nontemporal_load(q + (size_t)rr * row_stride16 + c16);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc ^= w[u].x ^ w[u].y ^ w[u].z ^ w[u].w;
  }
  if (acc == 0x12345678u) y[0] = 1.f;
}
template <int SEGL>
static void launch_tiled(const LaunchArgs& a, hipStream_t stream, int target_wgs) {
  const int rows_total = a.ga.seg[0].gm.units_total * (a.ga.seg[0].gm.K / a.ga.seg[0].gm.units_total == 8 ? 1 : 3);
  const int row_stride16 = a.ga.seg[0].gm.N / 4;
  const int col_tiles = (row_stride16 + SEGL - 1) / SEGL;
  int slices = (target_wgs + col_tiles - 1) / col_tiles;
  if (slices < 1) slices = 1;
  int rows_per_wg = (rows_total + slices - 1) / slices;
  const int gran = 16 * (64 / SEGL);
  rows_per_wg = (rows_per_wg + gran - 1) / gran * gran;
  slices = (rows_total + rows_per_wg - 1) / rows_per_wg;
  hipExtLaunchKernelGGL((sqllm_calib_tiled<SEGL>), dim3(col_tiles * slices), dim3(256), 0, stream, a.ev_start, a.ev_stop, 0,
                        reinterpret_cast<const u32x4*>(a.ga.seg[0].q), rows_total, row_stride16, col_tiles, rows_per_wg, a.ga.seg[0].y);
}
static hipError_t launch_calib(const LaunchArgs& a, hipStream_t stream) {
  if (a.ablate >= 200) {  // 2SW: S = log2(lanes per segment) - 3 (0..3 -> 8,16,32,64 lanes), W = target wgs / 256
    const int sg = (a.ablate / 10) % 10, tw = (a.ablate % 10) * 256;
    if (sg == 0) launch_tiled<8>(a, stream, tw);
    else if (sg == 1) launch_tiled<16>(a, stream, tw);
    else if (sg == 2) launch_tiled<32>(a, stream, tw);
    else launch_tiled<64>(a, stream, tw);
    return hipGetLastError();
  }
  const size_t n16 = (size_t)a.ga.seg[0].gm.units_total * (a.ga.seg[0].gm.K / a.ga.seg[0].gm.units_total == 8 ? 1 : 3) * (a.ga.seg[0].gm.N / 4);
  const int mode = a.ablate;
  dim3 grid(mode == 100 ? 512 : (mode % 10 == 1 ? 512 : mode % 10 == 2 ? 1024 : mode % 10 == 3 ? 2048 : 4096));
  if (mode == 100) hipExtLaunchKernelGGL(sqllm_calib_empty, grid, dim3(256), 0, stream, a.ev_start, a.ev_stop, 0, a.ga.seg[0].y);
  else if (mode < 120) hipExtLaunchKernelGGL((sqllm_calib_stream<4, true>), grid, dim3(256), 0, stream, a.ev_start, a.ev_stop, 0, reinterpret_cast<const u32x4*>(a.ga.seg[0].q), n16, a.ga.seg[0].y);
  else if (mode < 130) hipExtLaunchKernelGGL((sqllm_calib_stream<8, true>), grid, dim3(256), 0, stream, a.ev_start, a.ev_stop, 0, reinterpret_cast<const u32x4*>(a.ga.seg[0].q), n16, a.ga.seg[0].y);
  else hipExtLaunchKernelGGL((sqllm_calib_stream<8, false>), grid, dim3(256), 0, stream, a.ev_start, a.ev_stop, 0, reinterpret_cast<const u32x4*>(a.ga.seg[0].q), n16, a.ga.seg[0].y);
  return hipGetLastError();
}

// true = the launch was one of mine (*err is its result)
static bool fused_variant(int bits, const LaunchArgs& a, hipStream_t stream, hipError_t* err) {
  if (a.ablate >= 100) { *err = launch_calib(a, stream); return true; }
  if (!a.linear && bits == 4 && batch_tile(a.ga.seg[0].gm.batch) == 1 && a.ablate) {
    switch (a.ablate) {
      case 1: *err = launch_inst<4, 1, kWaves, 1>(a, stream); return true;
      case 2: *err = launch_inst<4, 1, kWaves, 2>(a, stream); return true;
      case 4: *err = launch_inst<4, 1, kWaves, 4>(a, stream); return true;
      case 8: *err = launch_inst<4, 1, kWaves, 8>(a, stream); return true;
      case 13: *err = launch_inst<4, 1, kWaves, 13>(a, stream); return true;
      case 14: *err = launch_inst<4, 1, kWaves, 14>(a, stream); return true;
      case 16: *err = launch_inst<4, 1, kWaves, 16>(a, stream); return true;
      case 32: *err = launch_inst<4, 1, kWaves, 32>(a, stream); return true;
      case 40: *err = launch_inst<4, 1, kWaves, 128>(a, stream); return true;  // option value 40 = ABL bit 128
      default: break;
    }
  }
  return false;
}

static struct InstallFusedVariant {
  InstallFusedVariant() { g_fused_variant = fused_variant; }
} g_install_fused_variant;

}  // namespace sqllm
