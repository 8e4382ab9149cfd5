// Developer tool: does vector-ALU work hide behind v_mfma_f32_32x32x2_f32 on this card?
// Per MFMA (64 cycles of the matrix pipe) each wave issues V independent v_fma_f32
// (4 cycles of the SIMD's vector ALU each); 1, 2 or 4 waves per SIMD.  Time per MFMA
// in shader cycles per SIMD: 64 if the vector work hides completely, 64 + 4 V if the
// two pipes exclude each other.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o garage_amd/_C/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

typedef float f2v __attribute__((ext_vector_type(2)));
// the library's tanh (common.h ga_tanh), its two-at-a-time form, and a form on
// exp(-2|x|) without the overflow clamp
__device__ __forceinline__ float tanh_a(float x) {
  const float t = 2.f * x;
  const float e = __expf(t > 80.f ? 80.f : t);
  const float d = e + 1.f;
  float r = __builtin_amdgcn_rcpf(d);
  r = fmaf(fmaf(-d, r, 1.f), r, r);
  return fmaf(-2.f, r, 1.f);
}
__device__ __forceinline__ f2v tanh_b(f2v x) {
  f2v t = x * 2.f;
  t.x = t.x > 80.f ? 80.f : t.x;
  t.y = t.y > 80.f ? 80.f : t.y;
  t = t * 1.44269504088896341f;
  f2v e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
  const f2v d = e + 1.f;
  f2v r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  const f2v one = {1.f, 1.f};
  r = __builtin_elementwise_fma(__builtin_elementwise_fma(-d, r, one), r, r);
  const f2v m2 = {-2.f, -2.f};
  return __builtin_elementwise_fma(m2, r, one);
}
__device__ __forceinline__ float tanh_c(float x) {
  const float e = __builtin_amdgcn_exp2f(-2.885390081777927f * __builtin_fabsf(x));
  const float d = e + 1.f;
  float r = __builtin_amdgcn_rcpf(d);
  r = fmaf(fmaf(-d, r, 1.f), r, r);
  const float m = fmaf(-2.f * e, r, 1.f);
  return __builtin_copysignf(m, x);
}
template <int KIND>
__global__ __launch_bounds__(1024) void tanh_loop(int iters, float* out, long long* clk) {
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3f + i * 0.1f - 0.8f;
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
      if (KIND == 0) { v[k] = tanh_a(v[k]) + 0.25f; v[k + 1] = tanh_a(v[k + 1]) + 0.25f; }
      if (KIND == 1) { f2v r = tanh_b(f2v{v[k], v[k + 1]}); v[k] = r.x + 0.25f; v[k + 1] = r.y + 0.25f; }
      if (KIND == 2) { v[k] = tanh_c(v[k]) + 0.25f; v[k + 1] = tanh_c(v[k + 1]) + 0.25f; }
    }
  }
  __syncthreads();  // every wave of the workgroup is done
  const long long t1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
  if (s == 12345.f) out[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}
template <int KIND>
void run_tanh(const char* name, float* out, long long* clk) {
  const int iters = 2000;
  for (int wps = 1; wps <= 4; wps *= 2) {
    long long h = 0, best = 1ll << 60;
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL((tanh_loop<KIND>), dim3(256), dim3(256 * wps), 0, 0, iters, out, clk);
      hipDeviceSynchronize();
      hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
      if (h < best) best = h;
    }
    printf("%-14s %d waves/SIMD: %.2f shader cycles per tanh (+1 add) per SIMD\n", name, wps,
           (double)best / ((double)wps * iters * 16.0));
  }
}

// The cost of one instruction of each kind with the matrix pipe idle: OP 0 v_fma_f32,
// 1 v_pk_fma_f32, 2 v_exp_f32, 3 v_rcp_f32, 4 v_cndmask_b32, 5 ds_read_b128 (per CU)
template <int OP>
__global__ __launch_bounds__(1024) void op_loop(int iters, float* out, long long* clk) {
  __shared__ float sh[4096];
  typedef float f2 __attribute__((ext_vector_type(2)));
  float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
  f2 a2 = {a, b}, b2 = {b, a};
  const unsigned long long msk = __ballot(threadIdx.x & 1);
  const float sa = __builtin_amdgcn_readfirstlane(a);
  float v[16];
  f2 w[8];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3f + i;
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = f2{v[i], v[i + 8]};
  sh[threadIdx.x] = a;
  __syncthreads();
  const float* lp = sh + (threadIdx.x & 63) * 4;
  f4 l = {0.f, 0.f, 0.f, 0.f};
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 64; ++k) {
      if (OP == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[k % 16]) : "v"(a), "v"(b));
      if (OP == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(w[k % 8]) : "v"(a2), "v"(b2));
      if (OP == 2) asm volatile("v_exp_f32 %0, %1" : "=v"(v[k % 16]) : "v"(a));
      if (OP == 3) asm volatile("v_rcp_f32 %0, %1" : "=v"(v[k % 16]) : "v"(a));
      if (OP == 4) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(v[k % 16]) : "v"(a), "v"(b));
      if (OP == 5) asm volatile("ds_read_b128 %0, %1" : "=v"(l) : "v"((unsigned)(uintptr_t)lp));
      if (OP == 6) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(v[k % 16]), "v"(b) : "vcc");
      if (OP == 7) asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(v[k % 16]) : "v"(a), "v"(b), "s"(msk));
      if (OP == 8) asm volatile("v_min_f32 %0, %1, %2" : "=v"(v[k % 16]) : "v"(a), "v"(b));
      if (OP == 9) asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(v[k % 16]) : "v"(a), "v"(b), "v"(v[(k + 8) % 16]));
      if (OP == 10) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[k % 16]) : "v"(a), "v"(b));
      if (OP == 11) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[k % 16]) : "s"(sa), "v"(b));
      if (OP == 12) asm volatile("ds_read_b32 %0, %1" : "=v"(l.x) : "v"((unsigned)(uintptr_t)lp));
      if (OP == 13) asm volatile("ds_write_b32 %0, %1" : : "v"((unsigned)(uintptr_t)lp), "v"(a) : "memory");
      if (OP == 14) asm volatile("ds_write_b128 %0, %1" : : "v"((unsigned)(uintptr_t)lp), "v"(l) : "memory");
      if (OP == 15) asm volatile("v_mov_b32 %0, %1" : "=v"(v[k % 16]) : "v"(a));
      if (OP == 16) asm volatile("v_add_u32 %0, %1, %2" : "=v"(v[k % 16]) : "v"(a), "v"(b));
      if (OP == 17) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(l) : "v"(a), "v"(b));
      if (OP == 18) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(w[k % 8]) : "v"(a2), "v"(b2));
    }
    if (OP == 12 || OP == 13 || OP == 14) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (OP == 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __syncthreads();  // every wave of the workgroup is done
  const long long t1 = clock64();
  float s = l.x;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += w[i].x + w[i].y;
  if (s == 12345.f) out[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

template <int OP>
void run_op(const char* name, float* out, long long* clk) {
  const int iters = 2000;
  for (int wps = 1; wps <= 4; wps *= 2) {
    const int threads = 256 * wps;  // one workgroup per CU: wps waves on every SIMD
    long long h = 0, best = 1ll << 60;
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL((op_loop<OP>), dim3(256), dim3(threads), 0, 0, iters, out, clk);
      hipDeviceSynchronize();
      hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
      if (h < best) best = h;
    }
    printf("%-14s %d waves/SIMD: %.2f shader cycles per instruction per SIMD\n", name,
           wps, (double)best / ((double)wps * iters * 64.0));
  }
}

template <int V, int LDS>
__global__ __launch_bounds__(1024) void loop(int iters, float* out, long long* clk) {
  __shared__ float sh[4096];
  f32x16 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3f + i;
  sh[threadIdx.x] = a;
  __syncthreads();
  const float* lp = sh + (threadIdx.x & 63) * 4;
  f4 l = {0.f, 0.f, 0.f, 0.f};
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < V; ++k)
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[k % 16]) : "v"(a), "v"(b));
        if (LDS) {
          asm volatile("ds_read_b128 %0, %1" : "=v"(l) : "v"((unsigned)(uintptr_t)lp));
        }
      }
    if (LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __syncthreads();  // every wave of the workgroup is done
  const long long t1 = clock64();
  float s = l.x;
#pragma unroll
  for (int c = 0; c < 4; ++c) s += acc[c][0];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
  if (s == 12345.f) out[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

// the same with v_mfma_f32_32x32x16_bf16 (8 passes: 32 cycles of the matrix pipe)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int V>
__global__ __launch_bounds__(1024) void loop_bf16(int iters, float* out, long long* clk) {
  f32x16 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
  bf16x8 a8, b8;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(a + i); b8[i] = (__bf16)(b + i); }
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3f + i;
  __syncthreads();
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[c], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < V; ++k)
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[k % 16]) : "v"(a), "v"(b));
      }
  }
  __syncthreads();
  const long long t1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) s += acc[c][0];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
  if (s == 12345.f) out[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}
template <int V>
void run_bf16(float* out, long long* clk) {
  const int iters = 2000;
  for (int wps = 1; wps <= 4; wps *= 2) {
    long long h = 0, best = 1ll << 60;
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL((loop_bf16<V>), dim3(256), dim3(256 * wps), 0, 0, iters, out, clk);
      hipDeviceSynchronize();
      hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
      if (h < best) best = h;
    }
    printf("bf16 32x32x16 + %2d v_fma, %d waves/SIMD: %6.1f shader cycles per MFMA per SIMD "
           "(32 = hidden)\n", V, wps, (double)best / ((double)wps * iters * 32.0));
  }
}

template <int V, int LDS>
void run(float* out, long long* clk) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd *= 2) {
    const int threads = 256 * waves_per_simd;  // one workgroup per CU
    const int grid = 256;
    float best = 1e9f; long long h = 0;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL((loop<V, LDS>), dim3(grid), dim3(threads), 0, 0, iters, out, clk);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) { best = ms; hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost); }
    }
    // MFMAs one SIMD issued: waves_per_simd * iters * 32
    const double per_mfma = (double)h / ((double)waves_per_simd * iters * 32.0);
    printf("V %2d lds %d, %d waves/SIMD: %.3f ms, %6.1f shader cycles per MFMA per SIMD "
           "(64 = hidden, %d = exclusive)\n", V, LDS, waves_per_simd, best, per_mfma, 64 + 4 * V);
  }
}

int main() {
  float* out; long long* clk;
  hipMalloc(&out, 64); hipMalloc(&clk, 64);
  run_tanh<0>("tanh scalar", out, clk);
  run_tanh<1>("tanh packed", out, clk);
  run_tanh<2>("tanh exp(-2|x|)", out, clk);
  run_op<0>("v_fma_f32", out, clk);
  run_op<1>("v_pk_fma_f32", out, clk);
  run_op<2>("v_exp_f32", out, clk);
  run_op<3>("v_rcp_f32", out, clk);
  run_op<4>("v_cndmask_b32", out, clk);
  run_op<5>("ds_read_b128", out, clk);
  run_op<6>("v_cmp_gt_f32", out, clk);
  run_op<7>("v_cndmask sgpr", out, clk);
  run_op<8>("v_min_f32", out, clk);
  run_op<9>("v_med3_f32", out, clk);
  run_op<10>("v_mul_f32", out, clk);
  run_op<11>("v_fmac sgpr", out, clk);
  run_op<12>("ds_read_b32", out, clk);
  run_op<13>("ds_write_b32", out, clk);
  run_op<14>("ds_write_b128", out, clk);
  run_op<15>("v_mov_b32", out, clk);
  run_op<16>("v_add_u32", out, clk);
  run_op<17>("mfma_16x16x4", out, clk);
  run_op<18>("v_pk_mul_f32", out, clk);
  run_bf16<0>(out, clk);
  run_bf16<2>(out, clk);
  run_bf16<4>(out, clk);
  run_bf16<8>(out, clk);
  run_bf16<16>(out, clk);
  run<0, 0>(out, clk);
  run<2, 0>(out, clk);
  run<4, 0>(out, clk);
  run<8, 0>(out, clk);
  run<12, 0>(out, clk);
  run<16, 0>(out, clk);
  run<24, 0>(out, clk);
  run<0, 1>(out, clk);
  run<4, 1>(out, clk);
  run<8, 1>(out, clk);
  return 0;
}
