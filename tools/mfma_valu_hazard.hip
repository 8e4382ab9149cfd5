// Developer tool: do vector-ALU results of one wave change when OTHER waves of the same
// SIMD run v_mfma_f32_32x32x16_bf16 at the same time?  Half of the waves of every
// workgroup run a long chain of fp32 vector arithmetic (plain and packed FMAs / adds,
// LDS round trips) whose result depends on every instruction; the other half either
// idle (reference run) or issue bf16 (mode 1) / fp32 (mode 2) / f16 (mode 3) 32x32 MFMAs or
// bf16 16x16x32 MFMAs (mode 4) back to back.  The
// vector results of the three runs must be the same bits.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_hazard.hip -o garage_amd/_C/mfma_valu_hazard
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void hazard(int mode, int iters, float* out, float* sink, int victim) {
  __shared__ float sh[8][64 * 4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if ((wave & 1) && victim == 1) {
    // ---- checked waves running fp32 MFMAs: a chain whose result depends on every one
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.001f * (lane + r);
    float a = 0.5f + 0.001f * lane, b = 0.25f + 0.002f * (lane & 31);
    for (int it = 0; it < iters; ++it) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] *= 0.5f;
    }
    float* o = out + ((size_t)blockIdx.x * 512 + threadIdx.x) * 16;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = acc[r];
  } else if ((wave & 1) && victim >= 2) {
    // ---- checked waves running ONE packed-fp32 instruction form over and over
    f2 accv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) accv[i] = f2{0.001f * (lane + i), 0.002f * (lane + 2 * i)};
    f2 w = {0.99991f + 1e-6f * lane, 0.99993f}, nn = {1.0e-4f, 1.00003f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#define FORM(k, text) if (victim == k) asm volatile(text : "=v"(accv[i]) : "v"(accv[i]), "v"(nn), "v"(w));
        FORM(2, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]")
        FORM(3, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]")
        FORM(4, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]")
        FORM(5, "v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]")
        FORM(6, "v_pk_fma_f32 %0, %1, %2, %3")
        FORM(7, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]")
        FORM(8, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]")
        FORM(9, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1]")
        FORM(10, "v_pk_mul_f32 %0, %1, %2")
        FORM(11, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]")
        FORM(12, "v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]")
#undef FORM
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) { accv[i].x = accv[i].x * 0.5f + 0.25f; accv[i].y = accv[i].y * 0.5f + 0.25f; }
    }
    float* o = out + ((size_t)blockIdx.x * 512 + threadIdx.x) * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[2 * i] = accv[i].x; o[2 * i + 1] = accv[i].y; }
  } else if (wave & 1) {
    // ---- the vector waves
    float x[8];
    f2 p[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = 0.001f * (threadIdx.x + 7 * i + 13 * blockIdx.x) + 0.5f;
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = f2{x[2 * i], x[2 * i + 1]};
    const f2 a = {0.99993f, 1.00007f}, b = {1e-4f, -1e-4f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = fmaf(x[i], 0.99991f, 1.1e-4f * (i + 1));
#pragma unroll
      for (int i = 0; i < 4; ++i) p[i] = __builtin_elementwise_fma(p[i], a, b);
      // an LDS round trip as in a reduction: write a float4, read a neighbour's
      float4 v = make_float4(x[0], x[1], p[0].x, p[0].y);
      *reinterpret_cast<float4*>(&sh[wave][4 * lane]) = v;
      __builtin_amdgcn_wave_barrier();
      const float4 w = *reinterpret_cast<const float4*>(&sh[wave][4 * (lane ^ 1)]);
      x[2] += 1e-3f * w.x; x[3] += 1e-3f * w.y; p[1].x += 1e-3f * w.z; p[1].y += 1e-3f * w.w;
      __builtin_amdgcn_wave_barrier();
    }
    float* o = out + ((size_t)blockIdx.x * 512 + threadIdx.x) * 16;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = x[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[8 + 2 * i] = p[i].x; o[9 + 2 * i] = p[i].y; }
  } else if (mode != 0) {
    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    bf16x8 a8, b8;
    f16x8 h8a, h8b;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      a8[i] = (__bf16)(0.01f * (lane + i)); b8[i] = (__bf16)(0.02f * i);
      h8a[i] = (_Float16)(0.01f * (lane + i)); h8b[i] = (_Float16)(0.02f * i);
    }
    const float af = 0.01f * lane, bf = 0.5f;
    // (about as long as the vector waves' loop)
    for (int it = 0; it < iters * 3; ++it) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (mode == 1)
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[c], 0, 0, 0);
        else if (mode == 2)
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[c], 0, 0, 0);
        else if (mode == 3)
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8a, h8b, acc[c], 0, 0, 0);
        else {
          f4v t = {acc[c][0], acc[c][1], acc[c][2], acc[c][3]};
          t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, t, 0, 0, 0);
          acc[c][0] = t[0]; acc[c][1] = t[1]; acc[c][2] = t[2]; acc[c][3] = t[3];
        }
      }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) s += acc[c][0];
    if (s == 12345.f) sink[0] = s;
  }
}

int main() {
  const int grid = 1024, iters = 4000;
  const size_t n = (size_t)grid * 512 * 16;
  float *out, *sink;
  hipMalloc(&out, n * sizeof(float));
  hipMalloc(&sink, 64);
  float* ref = (float*)malloc(n * sizeof(float));
  float* got = (float*)malloc(n * sizeof(float));
  hipMemset(out, 0, n * sizeof(float));
  for (int victim = 0; victim <= 12; ++victim) {
  static const char* names[] = {"vector arithmetic + LDS", "fp32 MFMAs", "v_pk_fma_f32 op_sel:[0,1,0]",
      "v_pk_fma_f32 op_sel_hi:[1,0,1]", "v_pk_fma_f32 op_sel:[1,0,0]", "v_pk_fma_f32 op_sel_hi:[0,1,1]",
      "v_pk_fma_f32 (no swizzle)", "v_pk_fma_f32 op_sel:[0,0,1]", "v_pk_mul_f32 op_sel:[0,1]",
      "v_pk_add_f32 op_sel:[0,1]", "v_pk_mul_f32 (no swizzle)", "v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1]",
      "v_pk_fma_f32 neg_lo/neg_hi on src0"};
  printf("checked waves run %s\n", names[victim]);
  hipMemset(out, 0, n * sizeof(float));
  hipLaunchKernelGGL(hazard, dim3(grid), dim3(512), 0, 0, 0, iters, out, sink, victim);
  hipDeviceSynchronize();
  hipMemcpy(ref, out, n * sizeof(float), hipMemcpyDeviceToHost);
  for (int mode = 0; mode <= 4; ++mode)
    for (int rep = 0; rep < (mode == 1 ? 2 : 1); ++rep) {
      hipMemset(out, 0, n * sizeof(float));
      hipLaunchKernelGGL(hazard, dim3(grid), dim3(512), 0, 0, mode, iters, out, sink, victim);
      hipDeviceSynchronize();
      hipMemcpy(got, out, n * sizeof(float), hipMemcpyDeviceToHost);
      size_t bad = 0, first = 0;
      for (size_t i = 0; i < n; ++i)
        if (memcmp(&got[i], &ref[i], 4) != 0) { if (!bad) first = i; ++bad; }
      printf("other waves %s, run %d: %zu of %zu vector results differ from the idle run",
             mode == 0 ? "idle" : mode == 1 ? "bf16 MFMA" : mode == 2 ? "fp32 MFMA" : mode == 3 ? "f16 MFMA 32x32x16" : "bf16 MFMA 16x16x32", rep, bad, n / 2);
      if (bad)
        printf(" (first: block %zu thread %zu value %zu: %.9g vs %.9g)", first / (512 * 16),
               (first / 16) % 512, first % 16, got[first], ref[first]);
      printf("\n");
    }
  }
  return 0;
}
