// Shared helpers for the gfx950 kernels of garage_amd.  CDNA4 only: wave = 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define GA_WAVE 64

// ---- error plumbing (thread-local message, negative return codes) ----------
extern "C" const char* ga_last_error(void);
void ga_set_error(const char* fmt, ...);

#define GA_OK 0
#define GA_ERR_ARG (-1)
#define GA_ERR_HIP (-2)

#define GA_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      ga_set_error(__VA_ARGS__);         \
      return GA_ERR_ARG;                 \
    }                                    \
  } while (0)

#define GA_CHECK_LAUNCH(name)                                          \
  do {                                                                 \
    hipError_t e__ = hipGetLastError();                                \
    if (e__ != hipSuccess) {                                           \
      ga_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return GA_ERR_HIP;                                               \
    }                                                                  \
  } while (0)

static inline bool ga_aligned16(const void* p) {
  return (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
}

static inline int64_t ga_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- device helpers --------------------------------------------------------
// XCD-aware workgroup order.  Workgroups are dealt to the 8 XCDs round-robin by
// their linear id, and each XCD has its own L2, so workgroups that read the same
// operand tile should have equal id % 8 (one L2 fetch instead of one per XCD) and
// close ids (co-resident).  Groups of `n_members` such workgroups are laid out as
//   id = (grp / 8) * 8 * n_members + member * 8 + grp % 8
// with the last n_groups % 8 groups in plain order.
__device__ __forceinline__ void ga_xcd_group(int id, int n_groups, int n_members,
                                             int* grp, int* member) {
  const int chunk = 8 * n_members;
  const int full = (n_groups / 8) * chunk;
  if (id < full) {
    const int c = id / chunk, r = id % chunk;
    *grp = c * 8 + (r & 7);
    *member = r >> 3;
  } else {
    const int r = id - full;
    *grp = (n_groups / 8) * 8 + r / n_members;
    *member = r % n_members;
  }
}

// tanh(x) = 1 - 2 / (exp(2x) + 1) on the hardware exp / rcp units, the ONE tanh of
// every kernel (forward, rollout and training paths must agree): v_exp, v_rcp and
// one Newton step on the reciprocal (8 VALU instructions; an IEEE division in its
// place costs 11 more, and the activation epilogues are VALU bound).  Absolute error
// <= ~2e-7 over the whole range; the exponent is capped so that exp stays finite
// (2 / (e^80 + 1) is 0 in fp32: the result is exactly 1 from x = 40 on), -1 at the
// other end, NaN in -> NaN out.
__device__ __forceinline__ float ga_tanh(float x) {
  const float t = 2.f * x;
  const float e = __expf(t > 80.f ? 80.f : t);  // (not fminf: a NaN must get through)
  const float d = e + 1.f;
  float r = __builtin_amdgcn_rcpf(d);
  r = fmaf(fmaf(-d, r, 1.f), r, r);
  return fmaf(-2.f, r, 1.f);
}

// The scalar std parameter of GaussianMLPBaseModule.forward
// (torch/modules/gaussian_mlp_module.py:165-181): clamp to [log min_std,
// log max_std], then 'exp' -> log std = p, or 'softplus' -> std =
// log(1 + exp(exp(p))).  `has_min` carries two flags: bit 0 = a lower clamp is
// present, bit 1 = softplus parameterisation.  Returns log std and
// *chain = d(log std) / d(parameter) (0 through an active clamp: torch.clamp
// passes no gradient outside the range).
__host__ __device__ inline float ga_log_std(float param, int has_min, float min_log_std,
                                            int has_max, float max_log_std,
                                            float* chain) {
  float p = param, c = 1.f;
  if ((has_min & 1) && p < min_log_std) { p = min_log_std; c = 0.f; }
  if (has_max && p > max_log_std) { p = max_log_std; c = 0.f; }
  if (has_min & 2) {
    const float e = expf(p);
    const float sp = logf(1.f + expf(e));  // the reference's .exp().exp().add(1).log()
    c *= e / ((1.f + expf(-e)) * sp);
    p = logf(sp);
  }
  if (chain) *chain = c;
  return p;
}

__device__ __forceinline__ float ga_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;  // valid in lane 0
}

__device__ __forceinline__ double ga_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

__device__ __forceinline__ float ga_wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_down(v, o, 64));
  return v;
}

// Block-wide sum of doubles for blockDim.x == 256 (4 waves); result in thread 0.
__device__ __forceinline__ double ga_block_sum_256(double v, double* smem4) {
  v = ga_wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) smem4[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) r = smem4[0] + smem4[1] + smem4[2] + smem4[3];
  __syncthreads();
  return r;
}
