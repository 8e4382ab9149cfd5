// Per-row loss arithmetic of the PPO / VPG update, shared by the kernels that
// compute the loss inside a GEMM epilogue (fused_train.hip).  Same formulas, in
// the same operation order, as the stand-alone loss kernels of losses.hip:
//   gaussian policy   PPO._compute_objective (torch/algos/ppo.py:96-132) /
//                     VPG._compute_objective (vpg.py:434-454) on an
//                     Independent(Normal(mean, exp(log_std))) with a scalar,
//                     clamped log-std (torch/modules/gaussian_mlp_module.py:158-192)
//   categorical       the same objectives on Categorical(logits = softmax(scores)),
//                     the convention of the reference's torch categorical policies
//                     (torch/policies/categorical_cnn_policy.py:138-139; Q15)
//   value function    GaussianMLPValueFunction.compute_loss
//                     (torch/value_functions/gaussian_mlp_value_function.py:81-98)
// Each returns the row's objective (or NLL) term, the second batch sum the
// finalize step needs, and d(loss)/d(head output) already divided by M.
#pragma once
#include "common.h"

namespace {

constexpr double LR_HALF_LOG_2PI = 0.91893853320467274178;

struct LossRowArgs {
  int kind;                  // 0 Gaussian policy, 1 value NLL, 2 categorical policy
  const float* actions;      // [*, lda] gathered through idx
  int64_t lda;
  const float* old_ll;       // gathered through idx (algo 0)
  const float* adv;          // gathered through idx
  const float* returns;      // gathered through idx (kind 1)
  const int32_t* idx;        // row m of the minibatch is sample idx[m] (null: m)
  const float* log_std;      // device scalar parameter (kinds 0, 1)
  int has_min, has_max;
  float min_log_std, max_log_std;
  int A;                     // head width
  int algo;                  // 0 PPO clipped surrogate, 1 VPG
  float clip;
  float ent_coeff;
  int ent_regularized, ent_softplus, ent_stop_grad;
  int double_softmax;
  float invM;
};

__device__ __forceinline__ float lr_softplus(float x) {
  return x > 20.f ? x : log1pf(expf(x));
}
__device__ __forceinline__ float lr_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// d obj / d ll and the objective from the likelihood (ppo.py:119-132; torch.min
// backward gives the smaller input the gradient and splits ties)
__device__ __forceinline__ void lr_surrogate(const LossRowArgs& a, float ll, float old_ll,
                                             float adv, float* obj, float* g) {
  if (a.algo == 1) {
    *obj = ll * adv;
    *g = adv;
    return;
  }
  const float ratio = expf(ll - old_ll);
  const float lo = 1.f - a.clip, hi = 1.f + a.clip;
  const float rc = fminf(fmaxf(ratio, lo), hi);
  const float s1 = ratio * adv, s2 = rc * adv;
  *obj = fminf(s1, s2);
  const float g1 = adv * ratio;
  const float g2 = (ratio >= lo && ratio <= hi) ? adv * ratio : 0.f;
  *g = (s1 < s2) ? g1 : ((s1 > s2) ? g2 : 0.5f * (g1 + g2));
}

// One row.  out[8]: the head outputs of the row; act[8] / adv / old_ll / ret: the
// row's sample (already gathered).  dout[8] <- d(loss)/d(out) (zero beyond A);
// returns the row's term of the first batch sum (objective, or NLL) and, through
// *second, of the second one (log-std gradient numerator; 0 for categorical).
__device__ __forceinline__ double lr_row(const LossRowArgs& a, float s, float inv_var,
                                         const float (&out)[8], const float (&act)[8],
                                         float adv, float old_ll, float ret,
                                         float (&dout)[8], double* second) {
#pragma unroll
  for (int j = 0; j < 8; ++j) dout[j] = 0.f;
  if (a.kind == 0) {
    const float lognorm = s + (float)LR_HALF_LOG_2PI;
    float ll = 0.f, q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < a.A) {
        const float d = act[j] - out[j];
        const float z = d * d * inv_var;
        q += z;
        ll += -0.5f * z - lognorm;
      }
    float obj, g;
    lr_surrogate(a, ll, old_ll, adv, &obj, &g);
    const float scale = -g * a.invM * inv_var;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < a.A) dout[j] = scale * (act[j] - out[j]);
    *second = (double)(-g * (q - (float)a.A));
    return (double)obj;
  }
  if (a.kind == 1) {
    const float d = ret - out[0];
    const float z = d * d * inv_var;
    dout[0] = -d * inv_var * a.invM;
    *second = (double)(1.f - z);
    return (double)(0.5f * z + s + (float)LR_HALF_LOG_2PI);
  }
  // categorical (losses.hip: ppo_categorical_loss_kernel)
  float mx = out[0];
#pragma unroll
  for (int j = 1; j < 8; ++j)
    if (j < a.A) mx = fmaxf(mx, out[j]);
  float den = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (j < a.A) den += expf(out[j] - mx);
  float lse;
  if (!a.double_softmax) {
    lse = mx + logf(den);
  } else {
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < a.A) s2 += expf(expf(out[j] - mx) / den);
    lse = logf(s2);
  }
  const int cls = (int)act[0];
  float pr[8], lp[8];
  float ll = 0.f, H = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    pr[j] = 0.f; lp[j] = 0.f;
    if (j < a.A) {
      pr[j] = expf(out[j] - mx) / den;
      lp[j] = (a.double_softmax ? pr[j] : out[j]) - lse;
      H -= expf(lp[j]) * lp[j];
      if (j == cls) ll = lp[j];
    }
  }
  float Hs = H, dHs = 1.f;
  if (a.ent_softplus) {
    dHs = lr_sigmoid(H);
    Hs = lr_softplus(H);
  }
  float obj, g;
  lr_surrogate(a, ll, old_ll, adv, &obj, &g);
  if (a.ent_regularized) obj += a.ent_coeff * Hs;
  const float cH = (a.ent_regularized && !a.ent_stop_grad) ? a.ent_coeff * dHs : 0.f;
  float dp[8];
  float dot = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    dp[j] = 0.f;
    if (j < a.A) {
      const float q = expf(lp[j]);
      dp[j] = g * ((j == cls ? 1.f : 0.f) - q) - cH * q * (lp[j] + H);
      dot += dp[j] * pr[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (j < a.A)
      dout[j] = a.double_softmax ? -(pr[j] * (dp[j] - dot)) * a.invM : -dp[j] * a.invM;
  *second = (double)Hs;
  return (double)obj;
}

// The batch scalars from the batch sums (the finalize step of the loss kernels):
// loss value and d(loss)/d(log_std) (0 where the clamp is active or the kind has no
// log-std parameter).
__device__ __forceinline__ void lr_finish(const LossRowArgs& a, double first, double second,
                                          int64_t M, float* loss, float* dlogstd) {
  if (a.kind == 2) {
    *loss = (float)(-(first / (double)M));
    *dlogstd = 0.f;
    return;
  }
  float s = *a.log_std;
  if (a.kind == 1) {
    *loss = (float)(first / (double)M);
    *dlogstd = (float)(second / (double)M);
    return;
  }
  float chain;
  s = ga_log_std(s, a.has_min, a.min_log_std, a.has_max, a.max_log_std, &chain);
  double mean_obj = first / (double)M;
  double dls = second / (double)M;
  if (a.ent_regularized) {
    float ent = (float)a.A * (0.5f + (float)LR_HALF_LOG_2PI + s);
    float dent = (float)a.A;
    if (a.ent_softplus) {
      dent *= lr_sigmoid(ent);
      ent = lr_softplus(ent);
    }
    mean_obj += (double)(a.ent_coeff * ent);
    if (!a.ent_stop_grad) dls += -(double)(a.ent_coeff * dent);
  }
  *loss = (float)(-mean_obj);
  *dlogstd = chain != 0.f ? (float)dls * chain : 0.f;
}

}  // namespace
