// Developer tool: every workgroup of a launch reads the same [256 rows x 128 B]
// slice of a row-major weight matrix at the same time (the B tile of a k-step).
// Does the row stride matter (L2 channel conflicts)?
//   hipcc --offload-arch=gfx950 -O3 tools/l2_stride.hip -o garage_amd/_C/l2_stride
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(512) void tile_reads(const float* W, int ld, int steps, int reps,
                                                  float* out) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r = 0; r < reps; ++r)
    for (int s = 0; s < steps; ++s) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = threadIdx.x + 512 * i;
        const int row = f >> 3, k = 4 * (f & 7);
        const float4 v = *reinterpret_cast<const float4*>(W + (long)row * ld + 32 * s + k);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      __syncthreads();
    }
  if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = acc.x;
}

int main() {
  float *W, *out;
  hipMalloc(&W, 256 * 512 * sizeof(float));
  hipMemset(W, 0, 256 * 512 * sizeof(float));
  hipMalloc(&out, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int lds[] = {256, 260, 264, 272, 288, 320};
  for (int grid = 256; grid <= 512; grid += 256)
    for (int li = 0; li < 6; ++li) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(tile_reads, dim3(grid), dim3(512), 0, 0, W, lds[li], 8, 100, out);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("grid %d, row stride %d floats: %.3f ms for 800 tile rounds -> %.2f us per 32-KB "
             "tile round, %.1f TB/s L2->CU\n", grid, lds[li], best, best * 1e3 / 800.0,
             (double)grid * 32768.0 * 800.0 / best / 1e9);
    }
  return 0;
}
