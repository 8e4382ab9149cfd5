// Developer tool: how does v_mfma_f32_32x32x16_bf16 round when it adds small products to a
// large accumulator?  Element (0, 0) of the tile: c = 1.0 and products that are fractions
// of ulp(1) = 2^-23.  (The same cases on v_mfma_f32_32x32x2_f32 for comparison.)
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rounding.hip -o garage_amd/_C/mfma_rounding
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// lane l supplies row / column l % 32 and k = 8 (l / 32) .. + 7; nk products of value p
// (as a * b with a = 2^-12, b = p * 2^12) at row 0 / column 0, c(0, 0) = cin
__global__ void k_bf16(float cin, float p, int nk, float* out) {
  const int lane = threadIdx.x;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    const int k = 8 * (lane / 32) + e;
    const bool on = (lane % 32) == 0 && k < nk;
    a[e] = (__bf16)(on ? ldexpf(1.f, -12) : 0.f);
    b[e] = (__bf16)(on ? p * ldexpf(1.f, 12) : 0.f);
  }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  if (lane == 0) c[0] = cin;  // D(0, 0) lives in lane 0, register 0
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  if (lane == 0) out[0] = c[0];
}
__global__ void k_f32(float cin, float p, int nk, float* out) {
  const int lane = threadIdx.x;
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  if (lane == 0) c[0] = cin;
  // 32x32x2: lane supplies k = lane / 32; nk products in nk / 2 (+1) instructions
  for (int s = 0; 2 * s < nk; ++s) {
    const int k = 2 * s + lane / 32;
    const bool on = (lane % 32) == 0 && k < nk;
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(on ? ldexpf(1.f, -12) : 0.f,
                                             on ? p * ldexpf(1.f, 12) : 0.f, c, 0, 0, 0);
  }
  if (lane == 0) out[0] = c[0];
}

int main() {
  float* out;
  hipMalloc(&out, 64);
  const float ulp = ldexpf(1.f, -23);
  struct { float c, frac; int nk; const char* what; } cases[] = {
      {1.f, 0.75f, 1, "one product of 0.75 ulp"},
      {1.f, 0.25f, 1, "one product of 0.25 ulp"},
      {1.f, 0.5f, 1, "one product of 0.5 ulp (tie: even = 1.0)"},
      {1.f + ulp, 0.5f, 1, "c = 1 + ulp, one product of 0.5 ulp (tie: even = 1 + 2 ulp)"},
      {1.f, 0.125f, 16, "16 products of 0.125 ulp (sum 2 ulp)"},
      {1.f, 0.09375f, 16, "16 products of 0.09375 ulp (sum 1.5 ulp: tie)"},
      {1.f, 0.046875f, 16, "16 products of 0.046875 ulp (sum 0.75 ulp)"},
      {-1.f, -0.75f, 1, "c = -1, one product of -0.75 ulp"},
      {1.f, -0.75f, 1, "c = 1, one product of -0.75 ulp (1 - 0.75 ulp: nearest 1 - ulp... in ulp(1)/2 units)"},
  };
  for (auto& t : cases) {
    float h[2];
    hipLaunchKernelGGL(k_bf16, dim3(1), dim3(64), 0, 0, t.c, t.frac * ulp, t.nk, out);
    hipMemcpy(&h[0], out, 4, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k_f32, dim3(1), dim3(64), 0, 0, t.c, t.frac * ulp, t.nk, out);
    hipMemcpy(&h[1], out, 4, hipMemcpyDeviceToHost);
    const double exact = (double)t.c + (double)t.frac * ulp * t.nk;
    printf("%-78s exact %+.3f ulp  bf16 MFMA %+.2f ulp  fp32 MFMA %+.2f ulp\n", t.what,
           (exact - t.c) / ulp, ((double)h[0] - t.c) / ulp, ((double)h[1] - t.c) / ulp);
  }
  return 0;
}
