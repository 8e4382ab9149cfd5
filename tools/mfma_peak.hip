// Developer tool: what the fp32 matrix pipe of this card delivers with nothing else
// in the way -- register-only v_mfma_f32_32x32x2_f32 loops, 2 or 4 independent
// accumulation chains per wave, 1..4 waves per SIMD -- and at which shader clock.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o garage_amd/_C/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(512) void mfma_loop(int iters, float* out, long long* clk) {
  f32x16 acc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
  const long long t0 = clock64();
  const long long w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c)
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  const long long t1 = clock64();
  const long long w1 = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0];
  if (s == 12345.f) out[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

int main() {
  float* out; long long* clk;
  hipMalloc(&out, 64); hipMalloc(&clk, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  for (int chains = 2; chains <= 4; chains += 2)
    for (int threads = 256; threads <= 512; threads += 256)
      for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu) {
        const int grid = 256 * wg_per_cu;
        float best = 1e9f; long long h[2] = {0, 0};
        for (int rep = 0; rep < 5; ++rep) {
          hipEventRecord(e0, 0);
          if (chains == 2) hipLaunchKernelGGL(mfma_loop<2>, dim3(grid), dim3(threads), 0, 0, iters, out, clk);
          else hipLaunchKernelGGL(mfma_loop<4>, dim3(grid), dim3(threads), 0, 0, iters, out, clk);
          hipEventRecord(e1, 0); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (ms < best) { best = ms; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost); }
        }
        const double flops = (double)grid * (threads / 64) * iters * 8.0 * chains * 4096.0;
        printf("chains %d, %d waves/WG, %d WG/CU (%d waves/SIMD): %.3f ms  %.1f TFLOP/s  "
               "shader clock %.2f GHz\n", chains, threads / 64, wg_per_cu,
               threads / 64 * wg_per_cu / 4 ? threads / 64 * wg_per_cu / 4 : 1, best,
               flops / best / 1e9, (double)h[0] / ((double)h[1] * 10.0));
      }
  return 0;
}
