// Fused loss / gradient-seed / optimiser kernels of the PPO update (gfx950).
//
//   ga_ppo_gaussian_loss_f32   PPO._compute_objective + VPG._compute_loss_with_adv
//                              (torch/algos/ppo.py:96-132, vpg.py:324-347,408-454)
//                              forward AND the gradient wrt the policy mean /
//                              scalar log-std, in one pass over the minibatch
//   ga_gaussian_nll_loss_f32   GaussianMLPValueFunction.compute_loss
//                              (torch/value_functions/gaussian_mlp_value_function.py:81-98)
//   ga_gaussian_kl_f32         VPG._compute_kl_constraint (vpg.py:381-406)
//   ga_reduce_slabs_f32 / ga_adam_step_f32
//                              OptimizerWrapper.step == torch.optim.Adam.step
//                              (torch/optimizers/optimizer_wrapper.py:53-63)
//   ga_stats_* / ga_adv_center_f32 / ga_sub_scalar_f32
//                              VPG._compute_advantage centring (vpg.py:371-377)
//
// All are HBM / latency bound elementwise passes: one thread per sample (rows
// are <= a few tens of floats), fp32 math, fp64 block partials reduced in a
// fixed order (bitwise reproducible run to run; no float atomics).
#include "common.h"

namespace {

constexpr int RED_BLOCKS = 512;  // upper bound on partial-producing blocks
constexpr double HALF_LOG_2PI = 0.91893853320467274178;

__device__ __forceinline__ float softplusf(float x) {
  // F.softplus with beta=1, threshold=20 (torch default)
  return x > 20.f ? x : log1pf(expf(x));
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// Multi-block reductions finish in the same launch: every block publishes its
// partial sums, then takes a ticket; the block that draws the last one sees all of
// them (release fence before the ticket, acquire fence after it, device-scope
// loads) and adds them in block order -- the same order, hence the same bits, as a
// separate one-wave finalize launch -- and leaves the ticket at 0 for the next
// launch on this workspace.  The ticket lives behind the partials
// (ga_reduction_workspace_doubles() counts it; the workspace starts zeroed).
__device__ __forceinline__ bool ga_take_last_ticket(unsigned* ticket) {
  __shared__ int is_last;
  if (threadIdx.x == 0) {
    __threadfence();  // this block's partials before its ticket
    const unsigned t = atomicAdd(ticket, 1u);
    is_last = (t == gridDim.x - 1);
    if (is_last) *ticket = 0;
  }
  __syncthreads();
  const bool last = is_last != 0;
  if (last) __threadfence();  // every block's partials after its ticket
  return last;
}
__device__ __forceinline__ double ga_peek(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Sum of partials[2 b + which] over the blocks, one wave, fixed order.
__device__ __forceinline__ double ga_sum_partials(const double* partials, int nblocks,
                                                  int which) {
  double v = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 64) v += ga_peek(partials + 2 * b + which);
  return ga_wave_sum(v);
}

struct PpoLossParams {
  const float* mean;      // [M, ldm] policy means of the minibatch rows
  int64_t ldm;
  const float* actions;   // [*, lda] gathered through idx (or row i when null)
  int64_t lda;
  const float* old_ll;    // gathered through idx
  const float* adv;       // gathered through idx
  const int32_t* idx;
  const float* log_std;   // device scalar parameter (unclamped)
  float min_log_std;      // lower clamp, applied when has_min
  int has_min;
  float max_log_std;
  int has_max;
  int64_t M;
  int A;
  int algo;               // 0 PPO clipped surrogate, 1 VPG (ll * adv),
                          // 2 TRPO unclipped surrogate (ratio * adv)
  float clip;
  float ent_coeff;        // added to the objective when ent_regularized
  int ent_regularized, ent_softplus, ent_stop_grad;
  float* dmean;           // optional [M, ldm]: dLoss/dmean (already / M)
  float* ll_out;          // optional [M]: new log-likelihoods
  double* partials;       // [gridDim.x][2]: sum objective, sum dLoss/dlog_std * M
  // the scalars of the batch (written by the last block to finish)
  float* loss_out;        // null: a finalize launch follows
  float* grad_slab0;
  int64_t slab_stride, n_splits;
  unsigned* ticket;
};

// Loss value and the log-std gradient slot from the batch sums (thread 0 of the
// finalize kernel, or of the loss kernel when it is the only block).
__device__ void ppo_gaussian_finish(double obj, double ds, const float* log_std,
                                    int has_min, float min_log_std, int has_max,
                                    float max_log_std, int64_t M, int A,
                                    float ent_coeff, int ent_regularized,
                                    int ent_softplus, int ent_stop_grad,
                                    float* loss_out, float* grad_slab0,
                                    int64_t slab_stride, int64_t n_splits) {
  float chain;
  const float s = ga_log_std(*log_std, has_min, min_log_std, has_max, max_log_std, &chain);
  double mean_obj = obj / (double)M;
  double dlogstd = ds / (double)M;  // d(-mean obj)/ds through the likelihood
  if (ent_regularized) {
    // Independent Normal entropy: A * (0.5 + 0.5 log 2pi + s), state independent
    float ent = (float)A * (0.5f + (float)HALF_LOG_2PI + s);
    float dent = (float)A;
    if (ent_softplus) {
      dent *= sigmoidf(ent);
      ent = softplusf(ent);
    }
    mean_obj += (double)(ent_coeff * ent);
    if (!ent_stop_grad) dlogstd += -(double)(ent_coeff * dent);
  }
  *loss_out = (float)(-mean_obj);
  if (grad_slab0) {
    // (through the clamp and the std parameterisation: 0 when the clamp is active)
    grad_slab0[0] = chain != 0.f ? (float)dlogstd * chain : 0.f;
    for (int64_t k = 1; k < n_splits; ++k) grad_slab0[k * slab_stride] = 0.f;
  }
}

__global__ __launch_bounds__(256) void ppo_gaussian_loss_kernel(PpoLossParams p) {
  __shared__ double red[4];
  const float s = ga_log_std(*p.log_std, p.has_min, p.min_log_std, p.has_max,
                             p.max_log_std, nullptr);
  const float inv_var = expf(-2.f * s);
  const float lognorm = s + (float)HALF_LOG_2PI;
  const float invM = 1.f / (float)p.M;

  double obj_sum = 0.0, ds_sum = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.M;
       i += (int64_t)gridDim.x * 256) {
    const int64_t src = p.idx ? (int64_t)p.idx[i] : i;
    const float* mu = p.mean + i * p.ldm;
    const float* a = p.actions + src * p.lda;
    float ll = 0.f, q = 0.f;  // q = sum (a-mu)^2 / var
    for (int j = 0; j < p.A; ++j) {
      const float d = a[j] - mu[j];
      const float z = d * d * inv_var;
      q += z;
      ll += -0.5f * z - lognorm;
    }
    if (p.ll_out) p.ll_out[i] = ll;
    const float adv = p.adv[src];
    float obj, g;  // g = d obj / d ll
    if (p.algo == 1) {
      obj = ll * adv;
      g = adv;
    } else if (p.algo == 2) {
      // torch/algos/trpo.py:113-117: likelihood ratio times advantage
      const float ratio = expf(ll - p.old_ll[src]);
      obj = ratio * adv;
      g = obj;
    } else {
      const float ratio = expf(ll - p.old_ll[src]);
      const float lo = 1.f - p.clip, hi = 1.f + p.clip;
      const float rc = fminf(fmaxf(ratio, lo), hi);
      const float s1 = ratio * adv, s2 = rc * adv;
      obj = fminf(s1, s2);
      const float g1 = adv * ratio;                                  // via surr
      const float g2 = (ratio >= lo && ratio <= hi) ? adv * ratio : 0.f;  // via clip
      // torch.min backward: the smaller input gets the gradient, ties split it
      g = (s1 < s2) ? g1 : ((s1 > s2) ? g2 : 0.5f * (g1 + g2));
    }
    obj_sum += (double)obj;
    if (p.dmean) {
      float* dm = p.dmean + i * p.ldm;
      const float scale = -g * invM * inv_var;
      for (int j = 0; j < p.A; ++j) dm[j] = scale * (a[j] - mu[j]);
    }
    // d ll / d s = sum_j ((a-mu)^2/var - 1)
    ds_sum += (double)(-g * (q - (float)p.A));
  }
  const double o = ga_block_sum_256(obj_sum, red);
  const double d = ga_block_sum_256(ds_sum, red);
  double bo = o, bd = d;  // batch sums (the only block: its own)
  if (gridDim.x > 1 || !p.loss_out) {
    if (threadIdx.x == 0) {
      p.partials[2 * blockIdx.x + 0] = o;
      p.partials[2 * blockIdx.x + 1] = d;
    }
    if (!p.loss_out || !ga_take_last_ticket(p.ticket)) return;
    if (threadIdx.x >= 64) return;
    bo = ga_sum_partials(p.partials, gridDim.x, 0);
    bd = ga_sum_partials(p.partials, gridDim.x, 1);
  }
  if (threadIdx.x == 0 && p.loss_out)
    ppo_gaussian_finish(bo, bd, p.log_std, p.has_min, p.min_log_std, p.has_max,
                        p.max_log_std, p.M, p.A, p.ent_coeff, p.ent_regularized,
                        p.ent_softplus, p.ent_stop_grad, p.loss_out, p.grad_slab0,
                        p.slab_stride, p.n_splits);
}

struct PpoFinalizeParams {
  const double* partials;
  int nblocks;
  const float* log_std;
  float min_log_std, max_log_std;
  int has_min, has_max;
  int64_t M;
  int A;
  float ent_coeff;
  int ent_regularized, ent_softplus, ent_stop_grad;
  float* loss_out;      // scalar
  float* grad_slab0;    // optional: slot that receives dLoss/dlog_std
  int64_t slab_stride;  // the same slot of slabs 1..n_splits-1 is zeroed
  int64_t n_splits;
};

__global__ void ppo_gaussian_finalize_kernel(PpoFinalizeParams p) {
  // one wave: lanes stride over the block partials, then a fixed-order wave sum
  double obj = 0.0, ds = 0.0;
  for (int b = threadIdx.x; b < p.nblocks; b += 64) {
    obj += p.partials[2 * b + 0];
    ds += p.partials[2 * b + 1];
  }
  obj = ga_wave_sum(obj);
  ds = ga_wave_sum(ds);
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  ppo_gaussian_finish(obj, ds, p.log_std, p.has_min, p.min_log_std, p.has_max,
                      p.max_log_std, p.M, p.A, p.ent_coeff, p.ent_regularized,
                      p.ent_softplus, p.ent_stop_grad, p.loss_out, p.grad_slab0,
                      p.slab_stride, p.n_splits);
}

// ---- categorical policy head ---------------------------------------------------
// The reference has no torch CategoricalMLPPolicy; its torch categorical (CNN)
// policies feed softmax(net(x)) to Categorical(logits=...)
// (torch/policies/categorical_cnn_policy.py:138-139).
// double_softmax = 1 keeps that convention, 0 treats the MLP output as logits.
struct CatLossParams {
  const float* scores;   // [M, lds] MLP outputs of the minibatch rows
  int64_t lds;
  const float* actions;  // [*, lda] column 0 holds the class id (as float)
  int64_t lda;
  const float* old_ll;
  const float* adv;
  const int32_t* idx;
  int64_t M;
  int A;
  int double_softmax;
  int algo;
  float clip;
  float ent_coeff;
  int ent_regularized, ent_softplus, ent_stop_grad;
  float* dscores;        // optional [M, lds]: dLoss/dscores (already / M)
  float* ll_out;         // optional [M]
  float* ent_out;        // optional [M]: per-row entropy (after softplus if set)
  double* partials;      // [gridDim.x][2]: sum objective, sum entropy
  // the scalars of the batch (written by the last block to finish)
  float* loss_out;       // null: a finalize launch follows
  double* ent_sum_out;
  float* grad_slab0;
  int64_t slab_stride, n_splits;
  unsigned* ticket;
};

// Final log-probabilities lp[j] of the row, its entropy, and (for the
// gradient) the softmax p of the raw scores.  A is small (number of actions).
__device__ __forceinline__ void cat_row(const float* sc, int A, int dbl, float* lse_out,
                                        float* mx_out, float* den_out) {
  float mx = sc[0];
  for (int j = 1; j < A; ++j) mx = fmaxf(mx, sc[j]);
  float den = 0.f;
  for (int j = 0; j < A; ++j) den += expf(sc[j] - mx);
  *mx_out = mx;
  *den_out = den;
  if (!dbl) {
    *lse_out = mx + logf(den);  // lp[j] = sc[j] - lse
  } else {
    // logits' = p in [0,1]: logsumexp without a shift is safe
    float s2 = 0.f;
    for (int j = 0; j < A; ++j) s2 += expf(expf(sc[j] - mx) / den);
    *lse_out = logf(s2);        // lp[j] = p[j] - lse
  }
}

__global__ __launch_bounds__(256) void ppo_categorical_loss_kernel(CatLossParams p) {
  __shared__ double red[4];
  const float invM = 1.f / (float)p.M;
  double obj_sum = 0.0, ent_sum = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.M;
       i += (int64_t)gridDim.x * 256) {
    const int64_t src = p.idx ? (int64_t)p.idx[i] : i;
    const float* sc = p.scores + i * p.lds;
    const int a = (int)p.actions[src * p.lda];
    float lse, mx, den;
    cat_row(sc, p.A, p.double_softmax, &lse, &mx, &den);
    // log-prob of the taken action and the entropy H = -sum q lp
    float ll = 0.f, H = 0.f;
    for (int j = 0; j < p.A; ++j) {
      const float pj = expf(sc[j] - mx) / den;
      const float lp = (p.double_softmax ? pj : sc[j]) - lse;
      const float q = expf(lp);
      H -= q * lp;
      if (j == a) ll = lp;
    }
    float Hs = H, dHs = 1.f;  // softplus(H) and its derivative
    if (p.ent_softplus) {
      dHs = sigmoidf(H);
      Hs = softplusf(H);
    }
    if (p.ll_out) p.ll_out[i] = ll;
    if (p.ent_out) p.ent_out[i] = Hs;
    const float adv = p.adv[src];
    float obj, g;
    if (p.algo == 1) {
      obj = ll * adv;
      g = adv;
    } else if (p.algo == 2) {
      // torch/algos/trpo.py:113-117: likelihood ratio times advantage
      const float ratio = expf(ll - p.old_ll[src]);
      obj = ratio * adv;
      g = obj;
    } else {
      const float ratio = expf(ll - p.old_ll[src]);
      const float lo = 1.f - p.clip, hi = 1.f + p.clip;
      const float rc = fminf(fmaxf(ratio, lo), hi);
      const float s1 = ratio * adv, s2 = rc * adv;
      obj = fminf(s1, s2);
      const float g1 = adv * ratio;
      const float g2 = (ratio >= lo && ratio <= hi) ? adv * ratio : 0.f;
      g = (s1 < s2) ? g1 : ((s1 > s2) ? g2 : 0.5f * (g1 + g2));
    }
    if (p.ent_regularized) obj += p.ent_coeff * Hs;
    obj_sum += (double)obj;
    ent_sum += (double)Hs;
    if (p.dscores) {
      // dObj/dlogit'_j = g (1[j=a] - q_j) + c_H * (-q_j (lp_j + H))
      const float cH = (p.ent_regularized && !p.ent_stop_grad)
                           ? p.ent_coeff * dHs : 0.f;
      float* ds = p.dscores + i * p.lds;
      if (!p.double_softmax) {
        for (int j = 0; j < p.A; ++j) {
          const float lp = sc[j] - lse;
          const float q = expf(lp);
          const float d = g * ((j == a ? 1.f : 0.f) - q) - cH * q * (lp + H);
          ds[j] = -d * invM;
        }
      } else {
        // chain through p = softmax(scores): dz_k = p_k (dp_k - sum_j dp_j p_j)
        float dot = 0.f;
        for (int j = 0; j < p.A; ++j) {
          const float pj = expf(sc[j] - mx) / den;
          const float lp = pj - lse;
          const float q = expf(lp);
          const float dp = g * ((j == a ? 1.f : 0.f) - q) - cH * q * (lp + H);
          dot += dp * pj;
        }
        for (int k = 0; k < p.A; ++k) {
          const float pk = expf(sc[k] - mx) / den;
          const float lp = pk - lse;
          const float q = expf(lp);
          const float dp = g * ((k == a ? 1.f : 0.f) - q) - cH * q * (lp + H);
          ds[k] = -(pk * (dp - dot)) * invM;
        }
      }
    }
  }
  const double o = ga_block_sum_256(obj_sum, red);
  const double e = ga_block_sum_256(ent_sum, red);
  double bo = o, be = e;  // batch sums (the only block: its own)
  if (gridDim.x > 1 || !p.loss_out) {
    if (threadIdx.x == 0) {
      p.partials[2 * blockIdx.x + 0] = o;
      p.partials[2 * blockIdx.x + 1] = e;
    }
    if (!p.loss_out || !ga_take_last_ticket(p.ticket)) return;
    if (threadIdx.x >= 64) return;
    bo = ga_sum_partials(p.partials, gridDim.x, 0);
    be = ga_sum_partials(p.partials, gridDim.x, 1);
  }
  if (threadIdx.x == 0) {
    *p.loss_out = (float)(-(bo / (double)p.M));
    if (p.ent_sum_out) *p.ent_sum_out = be;
    if (p.grad_slab0)  // the (unused) log-std slot of the flat layout stays 0
      for (int64_t k = 0; k < p.n_splits; ++k) p.grad_slab0[k * p.slab_stride] = 0.f;
  }
}

__global__ void ppo_categorical_finalize_kernel(const double* partials, int nblocks,
                                                int64_t M, float* loss_out,
                                                double* ent_sum_out, float* grad_slab0,
                                                int64_t slab_stride, int64_t n_splits) {
  double o = 0.0, e = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 64) {
    o += partials[2 * b];
    e += partials[2 * b + 1];
  }
  o = ga_wave_sum(o);
  e = ga_wave_sum(e);
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  *loss_out = (float)(-(o / (double)M));
  if (ent_sum_out) *ent_sum_out = e;
  if (grad_slab0)  // the (unused) log-std slot of the flat layout stays 0
    for (int64_t k = 0; k < n_splits; ++k) grad_slab0[k * slab_stride] = 0.f;
}

// sum over rows of KL(old || new) for categorical heads
__global__ __launch_bounds__(256) void categorical_kl_kernel(
    const float* sc_old, const float* sc_new, int64_t ld, int64_t M, int A, int dbl,
    double* partials) {
  __shared__ double red[4];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < M;
       i += (int64_t)gridDim.x * 256) {
    const float* so = sc_old + i * ld;
    const float* sn = sc_new + i * ld;
    float lo, mo, den_o, ln, mn, dn;
    cat_row(so, A, dbl, &lo, &mo, &den_o);
    cat_row(sn, A, dbl, &ln, &mn, &dn);
    float kl = 0.f;
    for (int j = 0; j < A; ++j) {
      const float lpo = (dbl ? expf(so[j] - mo) / den_o : so[j]) - lo;
      const float lpn = (dbl ? expf(sn[j] - mn) / dn : sn[j]) - ln;
      kl += expf(lpo) * (lpo - lpn);
    }
    acc += (double)kl;
  }
  const double r = ga_block_sum_256(acc, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = r;
}

struct NllParams {
  const float* v;        // [M, ldv], value in column 0
  int64_t ldv;
  const float* returns;  // gathered through idx
  const int32_t* idx;
  const float* log_std;  // device scalar, no clamp (min_std=None, max_std=None)
  int64_t M;
  float* dv;             // optional [M, ldv] column 0
  double* partials;      // [gridDim.x][2]
  float* loss_out;       // written by the last block; null: a finalize launch follows
  float* grad_slab0;
  int64_t slab_stride, n_splits;
  unsigned* ticket;
};

__global__ __launch_bounds__(256) void gaussian_nll_kernel(NllParams p) {
  __shared__ double red[4];
  const float s = *p.log_std;
  const float inv_var = expf(-2.f * s);
  const float invM = 1.f / (float)p.M;
  double nll = 0.0, ds = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.M;
       i += (int64_t)gridDim.x * 256) {
    const int64_t src = p.idx ? (int64_t)p.idx[i] : i;
    const float d = p.returns[src] - p.v[i * p.ldv];
    const float z = d * d * inv_var;
    nll += (double)(0.5f * z + s + (float)HALF_LOG_2PI);
    ds += (double)(1.f - z);
    if (p.dv) p.dv[i * p.ldv] = -d * inv_var * invM;
  }
  const double a = ga_block_sum_256(nll, red);
  const double b = ga_block_sum_256(ds, red);
  double ba = a, bb = b;  // batch sums (the only block: its own)
  if (gridDim.x > 1 || !p.loss_out) {
    if (threadIdx.x == 0) {
      p.partials[2 * blockIdx.x + 0] = a;
      p.partials[2 * blockIdx.x + 1] = b;
    }
    if (!p.loss_out || !ga_take_last_ticket(p.ticket)) return;
    if (threadIdx.x >= 64) return;
    ba = ga_sum_partials(p.partials, gridDim.x, 0);
    bb = ga_sum_partials(p.partials, gridDim.x, 1);
  }
  if (threadIdx.x == 0 && p.loss_out) {
    *p.loss_out = (float)(ba / (double)p.M);
    if (p.grad_slab0) {
      p.grad_slab0[0] = (float)(bb / (double)p.M);
      for (int64_t k = 1; k < p.n_splits; ++k) p.grad_slab0[k * p.slab_stride] = 0.f;
    }
  }
}

__global__ void gaussian_nll_finalize_kernel(const double* partials, int nblocks,
                                             int64_t M, float* loss_out,
                                             float* grad_slab0, int64_t slab_stride,
                                             int64_t n_splits) {
  double a = 0.0, b = 0.0;
  for (int k = threadIdx.x; k < nblocks; k += 64) {
    a += partials[2 * k];
    b += partials[2 * k + 1];
  }
  a = ga_wave_sum(a);
  b = ga_wave_sum(b);
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  *loss_out = (float)(a / (double)M);
  if (grad_slab0) {
    grad_slab0[0] = (float)(b / (double)M);
    for (int64_t k = 1; k < n_splits; ++k) grad_slab0[k * slab_stride] = 0.f;
  }
}

// KL(old || new) of Independent Normals with scalar log-stds, summed over rows.
__global__ __launch_bounds__(256) void gaussian_kl_kernel(
    const float* mean_old, const float* mean_new, int64_t ld, int64_t M, int A,
    float s_old, float s_new, double* partials) {
  __shared__ double red[4];
  // torch kl_normal_normal: 0.5 * (var_ratio + t1 - 1 - log(var_ratio))
  const float ratio = expf(s_old - s_new);
  const float var_ratio = ratio * ratio;
  const float inv_std_new = expf(-s_new);
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < M;
       i += (int64_t)gridDim.x * 256) {
    float kl = 0.f;
    for (int j = 0; j < A; ++j) {
      const float t = (mean_old[i * ld + j] - mean_new[i * ld + j]) * inv_std_new;
      kl += 0.5f * (var_ratio + t * t - 1.f - logf(var_ratio));
    }
    acc += (double)kl;
  }
  const double r = ga_block_sum_256(acc, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = r;
}

__global__ void sum_partials_kernel(const double* partials, int nblocks, double* out) {
  double a = 0.0;
  for (int k = threadIdx.x; k < nblocks; k += 64) a += partials[k];
  a = ga_wave_sum(a);
  if (threadIdx.x == 0 && blockIdx.x == 0) *out = a;
}

// ---- optimiser --------------------------------------------------------------
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* slabs,
                                                           int64_t n_splits,
                                                           int64_t stride, int64_t n,
                                                           float* out, float scale) {
  __builtin_amdgcn_s_setprio(3);  // (see adam_kernel)
  // 64 float4 columns x 4 slab groups per block: 4x the workgroups of a
  // column-per-thread layout and 4 independent 16-B loads in flight per thread;
  // the groups are combined through LDS in a fixed order (deterministic).
  __shared__ float4 part[4][64];
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t i4 = ((int64_t)blockIdx.x * 64 + col) * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i4 < n) {  // n is a multiple of 4 (flat buffers are padded)
    int64_t s = grp;
    for (; s + 12 < n_splits; s += 16) {
      const float4 v0 = *reinterpret_cast<const float4*>(slabs + s * stride + i4);
      const float4 v1 =
          *reinterpret_cast<const float4*>(slabs + (s + 4) * stride + i4);
      const float4 v2 =
          *reinterpret_cast<const float4*>(slabs + (s + 8) * stride + i4);
      const float4 v3 =
          *reinterpret_cast<const float4*>(slabs + (s + 12) * stride + i4);
      acc.x += (v0.x + v1.x) + (v2.x + v3.x);
      acc.y += (v0.y + v1.y) + (v2.y + v3.y);
      acc.z += (v0.z + v1.z) + (v2.z + v3.z);
      acc.w += (v0.w + v1.w) + (v2.w + v3.w);
    }
    for (; s < n_splits; s += 4) {
      const float4 v = *reinterpret_cast<const float4*>(slabs + s * stride + i4);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  part[grp][col] = acc;
  __syncthreads();
  if (grp == 0 && i4 < n) {
    const float4 a = part[0][col], b = part[1][col], c = part[2][col],
                 d = part[3][col];
    float4 r;
    r.x = ((a.x + b.x) + (c.x + d.x)) * scale;
    r.y = ((a.y + b.y) + (c.y + d.y)) * scale;
    r.z = ((a.z + b.z) + (c.z + d.z)) * scale;
    r.w = ((a.w + b.w) + (c.w + d.w)) * scale;
    *reinterpret_cast<float4*>(out + i4) = r;
  }
}

struct AdamParams {
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;
  float lerp_w;        // 1 - beta1
  float beta2, one_minus_beta2;
  float neg_step_size; // -lr / (1 - beta1^t)
  float bc2_sqrt;      // sqrt(1 - beta2^t)
  float eps;
};

// One element of torch.optim.Adam's single-tensor step with fused multiply-add
// contraction switched off (HIP's __fmul_rn / __fadd_rn are plain operators, so
// the compiler would otherwise be free to contract differently in each kernel):
// the stand-alone kernel and the slab-reduction + Adam kernel produce the same
// bits, which are those of one rounding per torch operation.
__device__ __forceinline__ void adam_update(const AdamParams& a, float g, float& p,
                                            float& m, float& v) {
#pragma clang fp contract(off)
  const float diff = g - m;
  m = fmaf(a.lerp_w, diff, m);             // exp_avg.lerp_(grad, 1 - beta1): torch's
                                           // lerp is one fused multiply-add
  const float gg = (a.one_minus_beta2 * g) * g;
  v = v * a.beta2 + gg;                    // mul_(beta2).addcmul_(g, g, 1 - beta2)
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  const float num = a.neg_step_size * m;
  p = p + num / denom;                     // addcdiv_(exp_avg, denom, -step_size)
}

__global__ __launch_bounds__(256) void adam_kernel(AdamParams a) {
  // (a short launch on its chain's critical path: its waves go first beside the other
  // chain's GEMMs, see reduce_regions_adam_kernel)
  __builtin_amdgcn_s_setprio(3);
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  float p = a.p[i], m = a.m[i], v = a.v[i];
  adam_update(a, a.g[i], p, m, v);
  a.p[i] = p;
  a.m[i] = m;
  a.v[i] = v;
}

// ---------------------------------------------------------------------------
// Optimizers other than the default Adam (make_optimizer, _functions.py:25-65,
// builds any torch.optim class): elementwise steps over the flat parameter buffer
// with torch's single-tensor arithmetic, one rounding per torch operation.
//   kind 1  torch.optim.SGD      h = {lr, momentum, dampening, weight_decay};
//                                flag bit 0 nesterov; s1 = momentum buffer
//   kind 2  torch.optim.RMSprop  h = {lr, alpha, eps, weight_decay, momentum};
//                                flag bit 0 centered; s1 = square_avg, s2 = momentum
//                                buffer, s3 = grad_avg
//   kind 3  torch.optim.Adam / AdamW beyond the defaults
//                                h = {lr, beta1, beta2, eps, weight_decay};
//                                flag bit 0 amsgrad, bit 1 decoupled decay (AdamW);
//                                s1 = exp_avg, s2 = exp_avg_sq, s3 = max_exp_avg_sq
struct OptParams {
  float* p;
  const float* g;
  float* s1;
  float* s2;
  float* s3;
  int64_t n;
  int kind, flags, first;  // first: step == 1 (SGD initialises its buffer with g)
  float lr, h1, h2, h3, h4;
  float neg_step_size, bc2_sqrt;  // kind 3
};

__global__ __launch_bounds__(256) void optimizer_step_kernel(OptParams a) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  float p = a.p[i], g = a.g[i];
  if (a.kind == 1) {
    const float momentum = a.h1, dampening = a.h2, wd = a.h3;
    if (wd != 0.f) g = g + wd * p;                      // grad.add(param, alpha=wd)
    if (momentum != 0.f) {
      float buf;
      if (a.first) {
        buf = g;                                        // torch.clone(grad)
      } else {
        buf = a.s1[i] * momentum;                       // buf.mul_(momentum)
        buf = buf + (1.f - dampening) * g;              //    .add_(grad, alpha=1-damp)
      }
      a.s1[i] = buf;
      g = (a.flags & 1) ? g + momentum * buf : buf;     // nesterov
    }
    p = p + (-a.lr) * g;                                // param.add_(grad, alpha=-lr)
  } else if (a.kind == 2) {
    const float alpha = a.h1, eps = a.h2, wd = a.h3, momentum = a.h4;
    if (wd != 0.f) g = g + wd * p;
    float sq = a.s1[i] * alpha;                         // square_avg.mul_(alpha)
    sq = sq + ((1.f - alpha) * g) * g;                  //    .addcmul_(g, g, 1-alpha)
    a.s1[i] = sq;
    float avg;
    if (a.flags & 1) {
      float ga = a.s3[i];
      ga = fmaf(1.f - alpha, g - ga, ga);               // grad_avg.lerp_(g, 1-alpha)
      a.s3[i] = ga;
      avg = sqrtf(sq + (-1.f * ga) * ga);               // addcmul(ga, ga, -1).sqrt_()
    } else {
      avg = sqrtf(sq);
    }
    avg = avg + eps;
    if (momentum > 0.f) {
      float buf = a.s2[i] * momentum;                   // buf.mul_(momentum)
      buf = buf + g / avg;                              //    .addcdiv_(grad, avg)
      a.s2[i] = buf;
      p = p + (-a.lr) * buf;
    } else {
      p = p + (-a.lr) * (g / avg);                      // addcdiv_(grad, avg, -lr)
    }
  } else {
    const float beta1 = a.h1, beta2 = a.h2, eps = a.h3, wd = a.h4;
    if (a.flags & 2) {
      p = p * (1.f - a.lr * wd);                        // AdamW: param.mul_(1 - lr wd)
    } else if (wd != 0.f) {
      g = g + wd * p;
    }
    float m = a.s1[i], v = a.s2[i];
    m = fmaf(1.f - beta1, g - m, m);                    // exp_avg.lerp_(grad, 1-beta1)
    v = v * beta2 + ((1.f - beta2) * g) * g;
    a.s1[i] = m;
    a.s2[i] = v;
    float vv = v;
    if (a.flags & 1) {                                  // amsgrad
      vv = fmaxf(a.s3[i], v);
      a.s3[i] = vv;
    }
    const float denom = sqrtf(vv) / a.bc2_sqrt + eps;
    p = p + (a.neg_step_size * m) / denom;
  }
  a.p[i] = p;
}

// reduce_slabs_kernel + adam_kernel in one launch (single process: nothing has
// to happen between the slab sum and the optimizer step).  Same arithmetic and
// the same summation order as the two separate kernels.
__global__ __launch_bounds__(256) void reduce_adam_kernel(const float* slabs,
                                                          int64_t n_splits,
                                                          int64_t stride, int zero_slot0,
                                                          AdamParams a, float* grads) {
  __builtin_amdgcn_s_setprio(3);  // (see adam_kernel)
  __shared__ float4 part[4][64];
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t i4 = ((int64_t)blockIdx.x * 64 + col) * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i4 < a.n) {
    int64_t s = grp;
    for (; s + 12 < n_splits; s += 16) {
      const float4 v0 = *reinterpret_cast<const float4*>(slabs + s * stride + i4);
      const float4 v1 =
          *reinterpret_cast<const float4*>(slabs + (s + 4) * stride + i4);
      const float4 v2 =
          *reinterpret_cast<const float4*>(slabs + (s + 8) * stride + i4);
      const float4 v3 =
          *reinterpret_cast<const float4*>(slabs + (s + 12) * stride + i4);
      acc.x += (v0.x + v1.x) + (v2.x + v3.x);
      acc.y += (v0.y + v1.y) + (v2.y + v3.y);
      acc.z += (v0.z + v1.z) + (v2.z + v3.z);
      acc.w += (v0.w + v1.w) + (v2.w + v3.w);
    }
    for (; s < n_splits; s += 4) {
      const float4 v = *reinterpret_cast<const float4*>(slabs + s * stride + i4);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  part[grp][col] = acc;
  __syncthreads();
  if (grp != 0 || i4 >= a.n) return;
  const float4 pa = part[0][col], pb = part[1][col], pc = part[2][col],
               pd = part[3][col];
  float g[4] = {(pa.x + pb.x) + (pc.x + pd.x), (pa.y + pb.y) + (pc.y + pd.y),
                (pa.z + pb.z) + (pc.z + pd.z), (pa.w + pb.w) + (pc.w + pd.w)};
  if (zero_slot0 && i4 == 0) g[0] = 0.f;  // log-std slot of a fixed-std module
  *reinterpret_cast<float4*>(grads + i4) = make_float4(g[0], g[1], g[2], g[3]);
  float4 p4 = *reinterpret_cast<float4*>(a.p + i4);
  float4 m4 = *reinterpret_cast<float4*>(a.m + i4);
  float4 v4 = *reinterpret_cast<float4*>(a.v + i4);
  float pp[4] = {p4.x, p4.y, p4.z, p4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w},
        vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) adam_update(a, g[j], pp[j], mm[j], vv[j]);
  *reinterpret_cast<float4*>(a.p + i4) = make_float4(pp[0], pp[1], pp[2], pp[3]);
  *reinterpret_cast<float4*>(a.m + i4) = make_float4(mm[0], mm[1], mm[2], mm[3]);
  *reinterpret_cast<float4*>(a.v + i4) = make_float4(vv[0], vv[1], vv[2], vv[3]);
}

// ---- advantage statistics ---------------------------------------------------
// stats (device, double[4]): [0] sum, [1] count, [2] sum of squared deviations,
// [3] minimum.  Between the stages a multi-GPU caller all-reduces the slots.
template <int WHAT>  // 0: sum, 1: squared deviations from stats mean, 2: min
__global__ __launch_bounds__(256) void stats_partial_kernel(const float* x, int64_t n,
                                                            const double* stats,
                                                            double* partials) {
  __shared__ double red[4];
  double mean = 0.0;
  if (WHAT == 1) mean = stats[0] / stats[1];
  double acc = (WHAT == 2) ? 1.0e300 : 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * 256) {
    const double v = (double)x[i];
    if (WHAT == 0) acc += v;
    if (WHAT == 1) acc += (v - mean) * (v - mean);
    if (WHAT == 2) acc = fmin(acc, v);
  }
  if (WHAT == 2) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc = fmin(acc, __shfl_down(acc, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
      partials[blockIdx.x] = fmin(fmin(red[0], red[1]), fmin(red[2], red[3]));
  } else {
    const double r = ga_block_sum_256(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = r;
  }
}

template <int WHAT>
__global__ void stats_finalize_kernel(const double* partials, int nblocks, int64_t n,
                                      double* stats) {
  if (WHAT == 2) {
    double a = 1.0e300;
    for (int k = threadIdx.x; k < nblocks; k += 64) a = fmin(a, partials[k]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a = fmin(a, __shfl_down(a, o, 64));
    if (threadIdx.x == 0 && blockIdx.x == 0) stats[3] = a;
  } else {
    double a = 0.0;
    for (int k = threadIdx.x; k < nblocks; k += 64) a += partials[k];
    a = ga_wave_sum(a);
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (WHAT == 0) {
      stats[0] = a;
      stats[1] = (double)n;
    } else {
      stats[2] = a;
    }
  }
}

__global__ __launch_bounds__(256) void adv_center_kernel(float* x, int64_t n,
                                                         const double* stats,
                                                         float eps) {
  // (a - mean) / (var + 1e-8), unbiased variance, in fp32 like the reference
  const float mean = (float)(stats[0] / stats[1]);
  const float var = (float)(stats[2] / (stats[1] - 1.0));
  const float denom = var + eps;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * 256)
    x[i] = (x[i] - mean) / denom;
}

__global__ __launch_bounds__(256) void sub_scalar_kernel(float* x, int64_t n,
                                                         const double* scalar) {
  const float s = (float)(*scalar);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * 256)
    x[i] -= s;
}

inline int red_blocks(int64_t n) {
  int64_t b = ga_ceil_div(n, 256);  // one row per thread up to RED_BLOCKS blocks
  if (b < 1) b = 1;
  if (b > RED_BLOCKS) b = RED_BLOCKS;
  return (int)b;
}

}  // namespace

// [RED_BLOCKS][2] partial sums + one slot for the ticket of the single-launch
// reductions / the grid-barrier words of small_step.hip (both leave it zero) + one
// slot whose first word small_step.hip raises when a barrier gave up
extern "C" int64_t ga_reduction_workspace_doubles(void) { return 2 * RED_BLOCKS + 2; }
extern "C" int64_t ga_reduction_partials_doubles(void) { return 2 * RED_BLOCKS; }

// A single block always finishes its own launch.  With several blocks the last
// ticket saves the finalize launch but pays two device-scope fences (L2 write-back
// / invalidate across the XCDs) in every block: measured +1 % per C3 iteration
// (146.5 vs 145.1 ms overlapped, 167.4 vs 166.1 ms on one stream) and +2 % at C2,
// so it is off by default.
static int g_one_launch_losses = 0;
extern "C" int ga_set_one_launch_losses(int on) {
  g_one_launch_losses = on != 0;
  return 0;
}

extern "C" int ga_ppo_gaussian_loss_f32(
    const float* mean, int64_t ldm, const float* actions, int64_t lda,
    const float* old_ll, const float* adv, const int32_t* idx, const float* log_std,
    int has_min, float min_log_std, int has_max, float max_log_std, int64_t M, int A,
    int algo, float clip, float ent_coeff, int ent_flags, float* dmean,
    float* ll_out, float* loss_out, float* grad_slab0, int64_t slab_stride,
    int64_t n_splits, double* workspace, hipStream_t stream) {
  GA_REQUIRE(mean && actions && adv && log_std && loss_out && workspace,
             "ga_ppo_gaussian_loss_f32: null pointer");
  GA_REQUIRE(algo == 1 || old_ll, "ga_ppo_gaussian_loss_f32: PPO needs old_ll");
  GA_REQUIRE(M > 0 && A > 0 && ldm >= A && lda >= A,
             "ga_ppo_gaussian_loss_f32: bad sizes");
  PpoLossParams p;
  p.mean = mean; p.ldm = ldm; p.actions = actions; p.lda = lda; p.old_ll = old_ll;
  p.adv = adv; p.idx = idx; p.log_std = log_std; p.min_log_std = min_log_std;
  p.has_min = has_min; p.max_log_std = max_log_std; p.has_max = has_max; p.M = M;
  p.A = A; p.algo = algo; p.clip = clip; p.ent_coeff = ent_coeff;
  p.ent_regularized = ent_flags & 1; p.ent_softplus = (ent_flags >> 1) & 1;
  p.ent_stop_grad = (ent_flags >> 2) & 1;
  p.dmean = dmean; p.ll_out = ll_out; p.partials = workspace;
  const int nb = red_blocks(M);
  const bool one_launch = nb == 1 || g_one_launch_losses != 0;
  p.loss_out = one_launch ? loss_out : nullptr;
  p.grad_slab0 = grad_slab0; p.slab_stride = slab_stride; p.n_splits = n_splits;
  p.ticket = reinterpret_cast<unsigned*>(workspace + 2 * RED_BLOCKS);
  hipLaunchKernelGGL(ppo_gaussian_loss_kernel, dim3(nb), dim3(256), 0, stream, p);
  GA_CHECK_LAUNCH("ppo_gaussian_loss");
  if (one_launch) return GA_OK;
  PpoFinalizeParams f;
  f.partials = workspace; f.nblocks = nb; f.log_std = log_std;
  f.min_log_std = min_log_std; f.max_log_std = max_log_std; f.has_min = has_min;
  f.has_max = has_max; f.M = M; f.A = A; f.ent_coeff = ent_coeff;
  f.ent_regularized = p.ent_regularized; f.ent_softplus = p.ent_softplus;
  f.ent_stop_grad = p.ent_stop_grad; f.loss_out = loss_out;
  f.grad_slab0 = grad_slab0; f.slab_stride = slab_stride; f.n_splits = n_splits;
  hipLaunchKernelGGL(ppo_gaussian_finalize_kernel, dim3(1), dim3(64), 0, stream, f);
  GA_CHECK_LAUNCH("ppo_gaussian_finalize");
  return GA_OK;
}

extern "C" int ga_ppo_categorical_loss_f32(
    const float* scores, int64_t lds, const float* actions, int64_t lda,
    const float* old_ll, const float* adv, const int32_t* idx, int64_t M, int A,
    int double_softmax, int algo, float clip, float ent_coeff, int ent_flags,
    float* dscores, float* ll_out, float* ent_out, float* loss_out,
    double* ent_sum_out, float* grad_slab0, int64_t slab_stride, int64_t n_splits,
    double* workspace, hipStream_t stream) {
  GA_REQUIRE(scores && actions && adv && loss_out && workspace,
             "ga_ppo_categorical_loss_f32: null pointer");
  GA_REQUIRE(algo == 1 || old_ll, "ga_ppo_categorical_loss_f32: PPO needs old_ll");
  GA_REQUIRE(M > 0 && A > 0 && lds >= A && lda >= 1,
             "ga_ppo_categorical_loss_f32: bad sizes");
  CatLossParams p;
  p.scores = scores; p.lds = lds; p.actions = actions; p.lda = lda;
  p.old_ll = old_ll; p.adv = adv; p.idx = idx; p.M = M; p.A = A;
  p.double_softmax = double_softmax; p.algo = algo; p.clip = clip;
  p.ent_coeff = ent_coeff; p.ent_regularized = ent_flags & 1;
  p.ent_softplus = (ent_flags >> 1) & 1; p.ent_stop_grad = (ent_flags >> 2) & 1;
  p.dscores = dscores; p.ll_out = ll_out; p.ent_out = ent_out;
  p.partials = workspace;
  const int nb = red_blocks(M);
  const bool one_launch = nb == 1 || g_one_launch_losses != 0;
  p.loss_out = one_launch ? loss_out : nullptr;
  p.ent_sum_out = ent_sum_out; p.grad_slab0 = grad_slab0;
  p.slab_stride = slab_stride; p.n_splits = n_splits;
  p.ticket = reinterpret_cast<unsigned*>(workspace + 2 * RED_BLOCKS);
  hipLaunchKernelGGL(ppo_categorical_loss_kernel, dim3(nb), dim3(256), 0, stream, p);
  GA_CHECK_LAUNCH("ppo_categorical_loss");
  if (one_launch) return GA_OK;
  hipLaunchKernelGGL(ppo_categorical_finalize_kernel, dim3(1), dim3(64), 0, stream,
                     (const double*)workspace, nb, M, loss_out, ent_sum_out,
                     grad_slab0, slab_stride, n_splits);
  GA_CHECK_LAUNCH("ppo_categorical_finalize");
  return GA_OK;
}

extern "C" int ga_categorical_kl_f32(const float* scores_old, const float* scores_new,
                                     int64_t ld, int64_t M, int A, int double_softmax,
                                     double* kl_sum_out, double* workspace,
                                     hipStream_t stream) {
  GA_REQUIRE(scores_old && scores_new && kl_sum_out && workspace,
             "ga_categorical_kl_f32: null pointer");
  GA_REQUIRE(M > 0 && A > 0 && ld >= A, "ga_categorical_kl_f32: bad sizes");
  const int nb = red_blocks(M);
  hipLaunchKernelGGL(categorical_kl_kernel, dim3(nb), dim3(256), 0, stream,
                     scores_old, scores_new, ld, M, A, double_softmax, workspace);
  GA_CHECK_LAUNCH("categorical_kl");
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, stream,
                     (const double*)workspace, nb, kl_sum_out);
  GA_CHECK_LAUNCH("sum_partials");
  return GA_OK;
}

extern "C" int ga_gaussian_nll_loss_f32(const float* v, int64_t ldv,
                                        const float* returns, const int32_t* idx,
                                        const float* log_std, int64_t M, float* dv,
                                        float* loss_out, float* grad_slab0,
                                        int64_t slab_stride, int64_t n_splits,
                                        double* workspace, hipStream_t stream) {
  GA_REQUIRE(v && returns && log_std && loss_out && workspace,
             "ga_gaussian_nll_loss_f32: null pointer");
  GA_REQUIRE(M > 0 && ldv >= 1, "ga_gaussian_nll_loss_f32: bad sizes");
  NllParams p;
  p.v = v; p.ldv = ldv; p.returns = returns; p.idx = idx; p.log_std = log_std;
  p.M = M; p.dv = dv; p.partials = workspace;
  const int nb = red_blocks(M);
  const bool one_launch = nb == 1 || g_one_launch_losses != 0;
  p.loss_out = one_launch ? loss_out : nullptr;
  p.grad_slab0 = grad_slab0; p.slab_stride = slab_stride; p.n_splits = n_splits;
  p.ticket = reinterpret_cast<unsigned*>(workspace + 2 * RED_BLOCKS);
  hipLaunchKernelGGL(gaussian_nll_kernel, dim3(nb), dim3(256), 0, stream, p);
  GA_CHECK_LAUNCH("gaussian_nll");
  if (one_launch) return GA_OK;
  hipLaunchKernelGGL(gaussian_nll_finalize_kernel, dim3(1), dim3(64), 0, stream,
                     (const double*)workspace, nb, M, loss_out, grad_slab0,
                     slab_stride, n_splits);
  GA_CHECK_LAUNCH("gaussian_nll_finalize");
  return GA_OK;
}

extern "C" int ga_gaussian_kl_f32(const float* mean_old, const float* mean_new,
                                  int64_t ld, int64_t M, int A, float log_std_old,
                                  float log_std_new, double* kl_sum_out,
                                  double* workspace, hipStream_t stream) {
  GA_REQUIRE(mean_old && mean_new && kl_sum_out && workspace,
             "ga_gaussian_kl_f32: null pointer");
  GA_REQUIRE(M > 0 && A > 0 && ld >= A, "ga_gaussian_kl_f32: bad sizes");
  const int nb = red_blocks(M);
  hipLaunchKernelGGL(gaussian_kl_kernel, dim3(nb), dim3(256), 0, stream, mean_old,
                     mean_new, ld, M, A, log_std_old, log_std_new, workspace);
  GA_CHECK_LAUNCH("gaussian_kl");
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, stream,
                     (const double*)workspace, nb, kl_sum_out);
  GA_CHECK_LAUNCH("sum_partials");
  return GA_OK;
}

extern "C" int ga_reduce_slabs_f32(const float* slabs, int64_t n_splits,
                                   int64_t slab_stride, int64_t n, float scale,
                                   float* out, hipStream_t stream) {
  GA_REQUIRE(slabs && out, "ga_reduce_slabs_f32: null pointer");
  GA_REQUIRE(n > 0 && n % 4 == 0 && slab_stride % 4 == 0 && n_splits >= 1,
             "ga_reduce_slabs_f32: n and stride must be multiples of 4");
  GA_REQUIRE(ga_aligned16(slabs) && ga_aligned16(out),
             "ga_reduce_slabs_f32: 16-B alignment required");
  const int64_t nb = ga_ceil_div(n / 4, 64);
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)nb), dim3(256), 0, stream,
                     slabs, n_splits, slab_stride, n, out, scale);
  GA_CHECK_LAUNCH("reduce_slabs");
  return GA_OK;
}

extern "C" int ga_reduce_adam_f32(const float* slabs, int64_t n_splits,
                                  int64_t slab_stride, float* params, float* grads,
                                  float* exp_avg, float* exp_avg_sq, int64_t n,
                                  int64_t step, double lr, double beta1, double beta2,
                                  double eps, int zero_slot0, hipStream_t stream) {
  GA_REQUIRE(slabs && params && grads && exp_avg && exp_avg_sq,
             "ga_reduce_adam_f32: null pointer");
  GA_REQUIRE(n > 0 && n % 4 == 0 && slab_stride % 4 == 0 && n_splits >= 1 && step >= 1,
             "ga_reduce_adam_f32: bad sizes");
  GA_REQUIRE(ga_aligned16(slabs) && ga_aligned16(params) && ga_aligned16(grads) &&
                 ga_aligned16(exp_avg) && ga_aligned16(exp_avg_sq),
             "ga_reduce_adam_f32: 16-B alignment required");
  AdamParams a;
  a.p = params; a.g = grads; a.m = exp_avg; a.v = exp_avg_sq; a.n = n;
  a.lerp_w = (float)(1.0 - beta1);
  a.beta2 = (float)beta2;
  a.one_minus_beta2 = (float)(1.0 - beta2);
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  a.neg_step_size = (float)(-(lr / bc1));
  a.bc2_sqrt = (float)sqrt(bc2);
  a.eps = (float)eps;
  const int64_t nb = ga_ceil_div(n / 4, 64);
  hipLaunchKernelGGL(reduce_adam_kernel, dim3((unsigned)nb), dim3(256), 0, stream,
                     slabs, n_splits, slab_stride, zero_slot0, a, grads);
  GA_CHECK_LAUNCH("reduce_adam");
  return GA_OK;
}

extern "C" int ga_adam_step_f32(float* params, const float* grads, float* exp_avg,
                                float* exp_avg_sq, int64_t n, int64_t step, double lr,
                                double beta1, double beta2, double eps,
                                hipStream_t stream) {
  GA_REQUIRE(params && grads && exp_avg && exp_avg_sq, "ga_adam_step_f32: null pointer");
  GA_REQUIRE(n > 0 && step >= 1, "ga_adam_step_f32: bad n / step");
  AdamParams a;
  a.p = params; a.g = grads; a.m = exp_avg; a.v = exp_avg_sq; a.n = n;
  a.lerp_w = (float)(1.0 - beta1);
  a.beta2 = (float)beta2;
  a.one_minus_beta2 = (float)(1.0 - beta2);
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  a.neg_step_size = (float)(-(lr / bc1));
  a.bc2_sqrt = (float)sqrt(bc2);
  a.eps = (float)eps;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)ga_ceil_div(n, 256)), dim3(256), 0,
                     stream, a);
  GA_CHECK_LAUNCH("adam");
  return GA_OK;
}

extern "C" int ga_optimizer_step_f32(int kind, float* params, const float* grads,
                                     float* s1, float* s2, float* s3, int64_t n,
                                     int64_t step, const double* h, int flags,
                                     hipStream_t stream) {
  GA_REQUIRE(params && grads && h, "ga_optimizer_step_f32: null pointer");
  GA_REQUIRE(n > 0 && step >= 1 && kind >= 1 && kind <= 3,
             "ga_optimizer_step_f32: bad n / step / kind");
  OptParams a;
  memset(&a, 0, sizeof(a));
  a.p = params; a.g = grads; a.s1 = s1; a.s2 = s2; a.s3 = s3; a.n = n;
  a.kind = kind; a.flags = flags; a.first = step == 1;
  a.lr = (float)h[0]; a.h1 = (float)h[1]; a.h2 = (float)h[2]; a.h3 = (float)h[3];
  a.h4 = (float)h[4];
  if (kind == 1) {
    GA_REQUIRE(h[1] == 0.0 || s1, "ga_optimizer_step_f32: SGD momentum buffer");
  } else if (kind == 2) {
    GA_REQUIRE(s1 && (h[4] <= 0.0 || s2) && (!(flags & 1) || s3),
               "ga_optimizer_step_f32: RMSprop state buffers");
  } else {
    GA_REQUIRE(s1 && s2 && (!(flags & 1) || s3),
               "ga_optimizer_step_f32: Adam state buffers");
    const double bc1 = 1.0 - pow(h[1], (double)step);
    const double bc2 = 1.0 - pow(h[2], (double)step);
    a.neg_step_size = (float)(-(h[0] / bc1));
    a.bc2_sqrt = (float)sqrt(bc2);
  }
  hipLaunchKernelGGL(optimizer_step_kernel, dim3((unsigned)ga_ceil_div(n, 256)),
                     dim3(256), 0, stream, a);
  GA_CHECK_LAUNCH("optimizer_step");
  return GA_OK;
}

extern "C" int ga_stats_f32(const float* x, int64_t n, int what, double* stats,
                            double* workspace, hipStream_t stream) {
  GA_REQUIRE(x && stats && workspace, "ga_stats_f32: null pointer");
  GA_REQUIRE(n > 0 && what >= 0 && what <= 2, "ga_stats_f32: bad arguments");
  const int nb = red_blocks(n);
  if (what == 0) {
    hipLaunchKernelGGL(stats_partial_kernel<0>, dim3(nb), dim3(256), 0, stream, x, n,
                       (const double*)stats, workspace);
    hipLaunchKernelGGL(stats_finalize_kernel<0>, dim3(1), dim3(64), 0, stream,
                       (const double*)workspace, nb, n, stats);
  } else if (what == 1) {
    hipLaunchKernelGGL(stats_partial_kernel<1>, dim3(nb), dim3(256), 0, stream, x, n,
                       (const double*)stats, workspace);
    hipLaunchKernelGGL(stats_finalize_kernel<1>, dim3(1), dim3(64), 0, stream,
                       (const double*)workspace, nb, n, stats);
  } else {
    hipLaunchKernelGGL(stats_partial_kernel<2>, dim3(nb), dim3(256), 0, stream, x, n,
                       (const double*)stats, workspace);
    hipLaunchKernelGGL(stats_finalize_kernel<2>, dim3(1), dim3(64), 0, stream,
                       (const double*)workspace, nb, n, stats);
  }
  GA_CHECK_LAUNCH("stats");
  return GA_OK;
}

extern "C" int ga_adv_center_f32(float* x, int64_t n, const double* stats, float eps,
                                 hipStream_t stream) {
  GA_REQUIRE(x && stats && n > 0, "ga_adv_center_f32: bad arguments");
  hipLaunchKernelGGL(adv_center_kernel, dim3(red_blocks(n)), dim3(256), 0, stream, x,
                     n, stats, eps);
  GA_CHECK_LAUNCH("adv_center");
  return GA_OK;
}

extern "C" int ga_sub_scalar_f32(float* x, int64_t n, const double* scalar,
                                 hipStream_t stream) {
  GA_REQUIRE(x && scalar && n > 0, "ga_sub_scalar_f32: bad arguments");
  hipLaunchKernelGGL(sub_scalar_kernel, dim3(red_blocks(n)), dim3(256), 0, stream, x,
                     n, scalar);
  GA_CHECK_LAUNCH("sub_scalar");
  return GA_OK;
}

// ---- vector ops of the constrained (TRPO) policy step ------------------------------
// torch/optimizers/conjugate_gradient_optimizer.py:69-104,146-186 works on the flat
// parameter vector (~72 k floats at C3): dot products and axpy-type updates.
namespace {

// one workgroup, fp64 accumulation in a fixed order: bitwise reproducible
__global__ __launch_bounds__(1024) void dot_kernel(const float* a, const float* b,
                                                   int64_t n, double* out) {
  __shared__ double part[16];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) s += (double)a[i] * (double)b[i];
  s = ga_wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += part[w];
    *out = t;
  }
}

// y = alpha * x + beta * y  (fp32, one fma rounding like torch's addcmul-free form)
__global__ __launch_bounds__(256) void axpby_kernel(float alpha, const float* x,
                                                    float beta, float* y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = alpha * x[i] + beta * y[i];
}

// Gaussian metric seed of the Fisher-vector product: with old == new, the Hessian
// of mean_i KL(old || new) with respect to the means is 1 / (sigma^2 M)
// (kl_normal_normal: 0.5 ((mu_o - mu)^2 / sigma^2 + ...)), so
//   dout[i, j] = tmean[i, j] * exp(-2 s) / M.
__global__ __launch_bounds__(256) void fisher_seed_kernel(
    const float* tmean, int64_t ldt, int64_t M, int A, const float* log_std,
    int has_min, float min_log_std, int has_max, float max_log_std, float* dout,
    int64_t ldd) {
  const float s = ga_log_std(*log_std, has_min, min_log_std, has_max, max_log_std, nullptr);
  const float scale = expf(-2.f * s) / (float)M;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M) return;
  for (int j = 0; j < (int)ldd; ++j)
    dout[i * ldd + j] = (j < A) ? tmean[i * ldt + j] * scale : 0.f;
}

// Categorical metric seed of the Fisher-vector product (TRPO with the categorical
// head).  With old == new the Hessian of mean_i KL(old || new) with respect to the
// LOGITS l of Categorical(logits=l) is (diag(q) - q q^T) / M, q = softmax(l)
// (the KL's gradient vanishes there, so no second-derivative term of the network
// enters: conjugate_gradient_optimizer.py:18-66 differentiates the same KL twice).
// The reference's head feeds l = softmax(scores) (categorical_cnn_policy.py:138-139,
// double_softmax), whose Jacobian J1 = diag(z) - z z^T is symmetric, so with the
// tangent t of the scores:
//   u = J1 t,  w = (diag(q) - q q^T) u,  dout = J1 w / M      (J1 = I without it).
// One thread per row, A <= 32 classes in registers.
constexpr int FS_MAX_A = 32;
__global__ __launch_bounds__(256) void fisher_seed_categorical_kernel(
    const float* scores, int64_t lds, const float* tscores, int64_t ldt, int64_t M, int A,
    int double_softmax, float* dout, int64_t ldd) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M) return;
  float z[FS_MAX_A], q[FS_MAX_A], u[FS_MAX_A];
  float mx = scores[i * lds];
  for (int j = 1; j < A; ++j) mx = fmaxf(mx, scores[i * lds + j]);
  float den = 0.f;
  for (int j = 0; j < A; ++j) {
    z[j] = expf(scores[i * lds + j] - mx);
    den += z[j];
  }
  for (int j = 0; j < A; ++j) z[j] /= den;
  if (double_softmax) {
    float den2 = 0.f;
    for (int j = 0; j < A; ++j) {
      q[j] = expf(z[j]);  // z in (0, 1]: no shift needed
      den2 += q[j];
    }
    for (int j = 0; j < A; ++j) q[j] /= den2;
    float zt = 0.f;
    for (int j = 0; j < A; ++j) zt += z[j] * tscores[i * ldt + j];
    for (int j = 0; j < A; ++j) u[j] = z[j] * (tscores[i * ldt + j] - zt);
  } else {
    for (int j = 0; j < A; ++j) {
      q[j] = z[j];
      u[j] = tscores[i * ldt + j];
    }
  }
  float qu = 0.f;
  for (int j = 0; j < A; ++j) qu += q[j] * u[j];
  for (int j = 0; j < A; ++j) u[j] = q[j] * (u[j] - qu);  // w
  const float inv_m = 1.f / (float)M;
  if (double_softmax) {
    float zw = 0.f;
    for (int j = 0; j < A; ++j) zw += z[j] * u[j];
    for (int j = 0; j < A; ++j) u[j] = z[j] * (u[j] - zw);
  }
  for (int j = 0; j < (int)ldd; ++j) dout[i * ldd + j] = (j < A) ? u[j] * inv_m : 0.f;
}

}  // namespace

extern "C" int ga_dot_f32(const float* a, const float* b, int64_t n, double* out,
                          hipStream_t stream) {
  GA_REQUIRE(a && b && out && n > 0, "ga_dot_f32: bad arguments");
  hipLaunchKernelGGL(dot_kernel, dim3(1), dim3(1024), 0, stream, a, b, n, out);
  GA_CHECK_LAUNCH("dot");
  return GA_OK;
}

extern "C" int ga_axpby_f32(double alpha, const float* x, double beta, float* y,
                            int64_t n, hipStream_t stream) {
  GA_REQUIRE(x && y && n > 0, "ga_axpby_f32: bad arguments");
  hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)ga_ceil_div(n, 256)), dim3(256), 0,
                     stream, (float)alpha, x, (float)beta, y, n);
  GA_CHECK_LAUNCH("axpby");
  return GA_OK;
}

extern "C" int ga_fisher_seed_gaussian_f32(const float* tmean, int64_t ldt, int64_t M,
                                           int A, const float* log_std, int has_min,
                                           float min_log_std, int has_max,
                                           float max_log_std, float* dout, int64_t ldd,
                                           hipStream_t stream) {
  GA_REQUIRE(tmean && log_std && dout && M > 0 && A > 0 && ldt >= A && ldd >= A,
             "ga_fisher_seed_gaussian_f32: bad arguments");
  hipLaunchKernelGGL(fisher_seed_kernel, dim3((unsigned)ga_ceil_div(M, 256)), dim3(256),
                     0, stream, tmean, ldt, M, A, log_std, has_min, min_log_std,
                     has_max, max_log_std, dout, ldd);
  GA_CHECK_LAUNCH("fisher_seed");
  return GA_OK;
}

extern "C" int ga_fisher_seed_categorical_f32(const float* scores, int64_t lds,
                                              const float* tscores, int64_t ldt,
                                              int64_t M, int A, int double_softmax,
                                              float* dout, int64_t ldd,
                                              hipStream_t stream) {
  GA_REQUIRE(scores && tscores && dout && M > 0 && A > 0 && A <= FS_MAX_A && lds >= A &&
                 ldt >= A && ldd >= A,
             "ga_fisher_seed_categorical_f32: bad arguments");
  hipLaunchKernelGGL(fisher_seed_categorical_kernel, dim3((unsigned)ga_ceil_div(M, 256)),
                     dim3(256), 0, stream, scores, lds, tscores, ldt, M, A, double_softmax,
                     dout, ldd);
  GA_CHECK_LAUNCH("fisher_seed_categorical");
  return GA_OK;
}

// ---- head layer fused into the loss ------------------------------------------------
// In a minibatch step the network's last linear layer (hidden -> action means, or
// hidden -> value) feeds nothing but the loss.  These kernels compute it in the
// loss kernel: 16 lanes own one row of the last hidden activations (16-B loads, a
// whole row coalesced), the head weights sit in LDS, the A dot products are
// reduced over the 16 lanes with xor shuffles, and the per-row loss / gradient
// seed is the same arithmetic as ppo_gaussian_loss_kernel / gaussian_nll_kernel.
// One pass over the [rows x hidden] matrix replaces the narrow GEMM launch, the
// round trip of its output and the loss launch.
namespace {

struct HeadLayer {
  const float* H;       // [M, ldh] last hidden activations (minibatch rows)
  int64_t ldh;
  const float* W;       // [A][ldw], k contiguous
  int64_t ldw;
  const float* bias;    // [A]
  int K;                // hidden width, a multiple of 64
  float* out;           // optional [M, ldo] head outputs
  int64_t ldo;
};

constexpr int HL_THREADS = 256;

__device__ __forceinline__ float group16_sum(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}

// Rows per 16-lane group and block iteration: all their loads are issued before
// any arithmetic (two dependent global-memory latencies per 64 rows: the gathered
// row ids, then everything else).
template <int KV>
struct HeadRows {
  static constexpr int R = KV <= 4 ? 4 : 2;
};

// head outputs of one row from its activations h[KV] -> rowbuf[0..A) (LDS)
template <int KV>
__device__ __forceinline__ void head_row(const HeadLayer& L, int A, const float* wlds,
                                         const float4 (&h)[KV], int64_t row, bool live,
                                         int gl, float* rowbuf) {
  for (int a = 0; a < A; ++a) {
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < KV; ++v) {
      const float4 w =
          *reinterpret_cast<const float4*>(wlds + a * L.K + 4 * gl + 64 * v);
      s = fmaf(h[v].x, w.x, s);
      s = fmaf(h[v].y, w.y, s);
      s = fmaf(h[v].z, w.z, s);
      s = fmaf(h[v].w, w.w, s);
    }
    s = group16_sum(s) + wlds[A * L.K + a];
    if (gl == 0) {
      rowbuf[a] = s;
      if (L.out && live) L.out[row * L.ldo + a] = s;
    }
  }
}

// W[A][K] and bias[A] -> LDS
__device__ __forceinline__ void stage_head(const HeadLayer& L, int A, float* wlds) {
  for (int i = threadIdx.x; i < A * L.K / 4; i += HL_THREADS) {
    const int a = (4 * i) / L.K, k = (4 * i) % L.K;
    *reinterpret_cast<float4*>(wlds + 4 * i) =
        *reinterpret_cast<const float4*>(L.W + (int64_t)a * L.ldw + k);
  }
  for (int a = threadIdx.x; a < A; a += HL_THREADS) wlds[A * L.K + a] = L.bias[a];
}

template <int KV>
__global__ __launch_bounds__(HL_THREADS) void head_ppo_gaussian_kernel(HeadLayer L,
                                                                       PpoLossParams p) {
  constexpr int R = HeadRows<KV>::R;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wlds = smem;                                  // [A][K] + bias[A]
  float* rows = smem + p.A * L.K + ((p.A + 3) & ~3);   // [16 groups][R][A]
  __shared__ double red[4];
  stage_head(L, p.A, wlds);
  __syncthreads();
  const float s = ga_log_std(*p.log_std, p.has_min, p.min_log_std, p.has_max,
                             p.max_log_std, nullptr);
  const float inv_var = expf(-2.f * s);
  const float lognorm = s + (float)HALF_LOG_2PI;
  const float invM = 1.f / (float)p.M;
  const int gl = threadIdx.x & 15, rg = threadIdx.x >> 4;
  double obj_sum = 0.0, ds_sum = 0.0;
  // whole 16 R-row chunks so that every lane of a wave stays in the shuffles
  const int64_t n_chunk = (p.M + 16 * R - 1) / (16 * R);
  for (int64_t c = blockIdx.x; c < n_chunk; c += gridDim.x) {
    int64_t irow[R], src[R];
    bool live[R];
    float4 h[R][KV];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      irow[r] = (c * R + r) * 16 + rg;
      live[r] = irow[r] < p.M;
      const int64_t row = live[r] ? irow[r] : p.M - 1;
      src[r] = p.idx ? (int64_t)p.idx[row] : row;
#pragma unroll
      for (int v = 0; v < KV; ++v)
        h[r][v] = *reinterpret_cast<const float4*>(L.H + row * L.ldh + 4 * gl + 64 * v);
    }
    float advr[R], oldr[R], actr[R][4];  // lane gl holds action dims gl, gl+16, ..
#pragma unroll
    for (int r = 0; r < R; ++r) {
      advr[r] = p.adv[src[r]];
      oldr[r] = (p.algo == 1) ? 0.f : p.old_ll[src[r]];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int j = gl + 16 * t;
        actr[r][t] = (j < p.A) ? p.actions[src[r] * p.lda + j] : 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float* rowbuf = rows + (rg * R + r) * p.A;
      const int64_t row = live[r] ? irow[r] : p.M - 1;
      head_row<KV>(L, p.A, wlds, h[r], row, live[r], gl, rowbuf);
      // the 16 lanes of a group are in one wave: lane 0's LDS writes become
      // visible to the group after the wave-level fence
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      float ll_part = 0.f, q_part = 0.f, dj[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int j = gl + 16 * t;
        dj[t] = 0.f;
        if (j < p.A) {
          const float d = actr[r][t] - rowbuf[j];
          const float z = d * d * inv_var;
          dj[t] = d;
          q_part += z;
          ll_part += -0.5f * z - lognorm;
        }
      }
      const float q = group16_sum(q_part);
      const float ll = group16_sum(ll_part);
      const float adv = advr[r];
      float obj, g;
      if (p.algo == 1) {
        obj = ll * adv;
        g = adv;
      } else if (p.algo == 2) {
        const float ratio = expf(ll - oldr[r]);
        obj = ratio * adv;
        g = obj;
      } else {
        const float ratio = expf(ll - oldr[r]);
        const float lo = 1.f - p.clip, hi = 1.f + p.clip;
        const float rc = fminf(fmaxf(ratio, lo), hi);
        const float s1 = ratio * adv, s2 = rc * adv;
        obj = fminf(s1, s2);
        const float g1 = adv * ratio;
        const float g2 = (ratio >= lo && ratio <= hi) ? adv * ratio : 0.f;
        g = (s1 < s2) ? g1 : ((s1 > s2) ? g2 : 0.5f * (g1 + g2));
      }
      if (live[r]) {
        if (p.ll_out && gl == 0) p.ll_out[irow[r]] = ll;
        if (p.dmean) {
          const float scale = -g * invM * inv_var;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int j = gl + 16 * t;
            if (j < p.A) p.dmean[irow[r] * p.ldm + j] = scale * dj[t];
          }
        }
        if (gl == 0) {
          obj_sum += (double)obj;
          ds_sum += (double)(-g * (q - (float)p.A));
        }
      }
    }
  }
  const double o = ga_block_sum_256(obj_sum, red);
  const double d = ga_block_sum_256(ds_sum, red);
  if (threadIdx.x == 0) {
    p.partials[2 * blockIdx.x + 0] = o;
    p.partials[2 * blockIdx.x + 1] = d;
  }
}

template <int KV>
__global__ __launch_bounds__(HL_THREADS) void head_gaussian_nll_kernel(HeadLayer L,
                                                                       NllParams p) {
  constexpr int R = HeadRows<KV>::R;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wlds = smem;                 // [1][K] + bias
  float* rows = smem + L.K + 4;       // [16 groups][R]
  __shared__ double red[4];
  stage_head(L, 1, wlds);
  __syncthreads();
  const float s = *p.log_std;
  const float inv_var = expf(-2.f * s);
  const float invM = 1.f / (float)p.M;
  const int gl = threadIdx.x & 15, rg = threadIdx.x >> 4;
  double nll = 0.0, ds = 0.0;
  const int64_t n_chunk = (p.M + 16 * R - 1) / (16 * R);
  for (int64_t c = blockIdx.x; c < n_chunk; c += gridDim.x) {
    int64_t irow[R];
    bool live[R];
    float4 h[R][KV];
    float retr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      irow[r] = (c * R + r) * 16 + rg;
      live[r] = irow[r] < p.M;
      const int64_t row = live[r] ? irow[r] : p.M - 1;
      const int64_t src = p.idx ? (int64_t)p.idx[row] : row;
#pragma unroll
      for (int v = 0; v < KV; ++v)
        h[r][v] = *reinterpret_cast<const float4*>(L.H + row * L.ldh + 4 * gl + 64 * v);
      retr[r] = p.returns[src];
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float* rowbuf = rows + rg * R + r;
      const int64_t row = live[r] ? irow[r] : p.M - 1;
      head_row<KV>(L, 1, wlds, h[r], row, live[r], gl, rowbuf);
      if (live[r] && gl == 0) {
        const float d = retr[r] - rowbuf[0];
        const float z = d * d * inv_var;
        nll += (double)(0.5f * z + s + (float)HALF_LOG_2PI);
        ds += (double)(1.f - z);
        if (p.dv) p.dv[irow[r] * p.ldv] = -d * inv_var * invM;
      }
    }
  }
  const double a = ga_block_sum_256(nll, red);
  const double b = ga_block_sum_256(ds, red);
  if (threadIdx.x == 0) {
    p.partials[2 * blockIdx.x + 0] = a;
    p.partials[2 * blockIdx.x + 1] = b;
  }
}

inline int head_blocks(int64_t M) {
  int64_t b = (M + 63) / 64;  // rows per workgroup
  if (b < 1) b = 1;
  if (b > RED_BLOCKS) b = RED_BLOCKS;
  return (int)b;
}

}  // namespace

extern "C" int ga_head_loss_supported(int hidden_width, int A) {
  return (hidden_width == 64 || hidden_width == 128 || hidden_width == 256 ||
          hidden_width == 512) && A >= 1 && A <= 64 &&
         (int64_t)A * hidden_width * 4 <= 60 * 1024;
}

extern "C" int ga_head_ppo_gaussian_loss_f32(
    const float* H, int64_t ldh, const float* W, int64_t ldw, const float* bias,
    int hidden_width, float* mean_out, int64_t ldm, const float* actions, int64_t lda,
    const float* old_ll, const float* adv, const int32_t* idx, const float* log_std,
    int has_min, float min_log_std, int has_max, float max_log_std, int64_t M, int A,
    int algo, float clip, float ent_coeff, int ent_flags, float* dmean, int64_t ldd,
    float* ll_out, float* loss_out, float* grad_slab0, int64_t slab_stride,
    int64_t n_splits, double* workspace, hipStream_t stream) {
  GA_REQUIRE(H && W && bias && actions && adv && log_std && loss_out && workspace,
             "ga_head_ppo_gaussian_loss_f32: null pointer");
  GA_REQUIRE(ga_head_loss_supported(hidden_width, A),
             "ga_head_ppo_gaussian_loss_f32: unsupported head %d x %d", A, hidden_width);
  GA_REQUIRE(algo == 1 || old_ll, "ga_head_ppo_gaussian_loss_f32: PPO needs old_ll");
  GA_REQUIRE(M > 0 && ldh >= hidden_width && ldh % 4 == 0 && ldw >= hidden_width &&
                 ldw % 4 == 0 && lda >= A && (!dmean || ldd >= A) &&
                 (!mean_out || ldm >= A) && ga_aligned16(H) && ga_aligned16(W),
             "ga_head_ppo_gaussian_loss_f32: bad sizes / alignment");
  HeadLayer L;
  L.H = H; L.ldh = ldh; L.W = W; L.ldw = ldw; L.bias = bias; L.K = hidden_width;
  L.out = mean_out; L.ldo = ldm;
  PpoLossParams p;
  p.mean = nullptr; p.ldm = ldd; p.actions = actions; p.lda = lda; p.old_ll = old_ll;
  p.adv = adv; p.idx = idx; p.log_std = log_std; p.min_log_std = min_log_std;
  p.has_min = has_min; p.max_log_std = max_log_std; p.has_max = has_max; p.M = M;
  p.A = A; p.algo = algo; p.clip = clip; p.ent_coeff = ent_coeff;
  p.ent_regularized = ent_flags & 1; p.ent_softplus = (ent_flags >> 1) & 1;
  p.ent_stop_grad = (ent_flags >> 2) & 1;
  p.dmean = dmean; p.ll_out = ll_out; p.partials = workspace;
  p.loss_out = nullptr; p.grad_slab0 = nullptr; p.slab_stride = 0; p.n_splits = 0;
  p.ticket = nullptr;
  const int nb = head_blocks(M);
  const unsigned lds =
      (unsigned)((A * hidden_width + ((A + 3) & ~3) + 16 * 4 * A) * sizeof(float));
  switch (hidden_width / 64) {
    case 1: hipLaunchKernelGGL(head_ppo_gaussian_kernel<1>, dim3(nb), dim3(HL_THREADS), lds, stream, L, p); break;
    case 2: hipLaunchKernelGGL(head_ppo_gaussian_kernel<2>, dim3(nb), dim3(HL_THREADS), lds, stream, L, p); break;
    case 4: hipLaunchKernelGGL(head_ppo_gaussian_kernel<4>, dim3(nb), dim3(HL_THREADS), lds, stream, L, p); break;
    default: hipLaunchKernelGGL(head_ppo_gaussian_kernel<8>, dim3(nb), dim3(HL_THREADS), lds, stream, L, p); break;
  }
  GA_CHECK_LAUNCH("head_ppo_gaussian");
  PpoFinalizeParams f;
  f.partials = workspace; f.nblocks = nb; f.log_std = log_std;
  f.min_log_std = min_log_std; f.max_log_std = max_log_std; f.has_min = has_min;
  f.has_max = has_max; f.M = M; f.A = A; f.ent_coeff = ent_coeff;
  f.ent_regularized = p.ent_regularized; f.ent_softplus = p.ent_softplus;
  f.ent_stop_grad = p.ent_stop_grad; f.loss_out = loss_out;
  f.grad_slab0 = grad_slab0; f.slab_stride = slab_stride; f.n_splits = n_splits;
  hipLaunchKernelGGL(ppo_gaussian_finalize_kernel, dim3(1), dim3(64), 0, stream, f);
  GA_CHECK_LAUNCH("ppo_gaussian_finalize");
  return GA_OK;
}

extern "C" int ga_head_gaussian_nll_loss_f32(
    const float* H, int64_t ldh, const float* W, const float* bias, int hidden_width,
    float* v_out, int64_t ldv_out, const float* returns, const int32_t* idx,
    const float* log_std, int64_t M, float* dv, int64_t ldd, float* loss_out,
    float* grad_slab0, int64_t slab_stride, int64_t n_splits, double* workspace,
    hipStream_t stream) {
  GA_REQUIRE(H && W && bias && returns && log_std && loss_out && workspace,
             "ga_head_gaussian_nll_loss_f32: null pointer");
  GA_REQUIRE(ga_head_loss_supported(hidden_width, 1),
             "ga_head_gaussian_nll_loss_f32: unsupported hidden width %d", hidden_width);
  GA_REQUIRE(M > 0 && ldh >= hidden_width && ldh % 4 == 0 && (!dv || ldd >= 1) &&
                 (!v_out || ldv_out >= 1) && ga_aligned16(H) && ga_aligned16(W),
             "ga_head_gaussian_nll_loss_f32: bad sizes / alignment");
  HeadLayer L;
  L.H = H; L.ldh = ldh; L.W = W; L.ldw = hidden_width; L.bias = bias;
  L.K = hidden_width; L.out = v_out; L.ldo = ldv_out;
  NllParams p;
  p.v = nullptr; p.ldv = ldd; p.returns = returns; p.idx = idx; p.log_std = log_std;
  p.M = M; p.dv = dv; p.partials = workspace;
  p.loss_out = nullptr; p.grad_slab0 = nullptr; p.slab_stride = 0; p.n_splits = 0;
  p.ticket = nullptr;
  const int nb = head_blocks(M);
  const unsigned lds = (unsigned)((hidden_width + 4 + 16 * 4) * sizeof(float));
  switch (hidden_width / 64) {
    case 1: hipLaunchKernelGGL(head_gaussian_nll_kernel<1>, dim3(nb), dim3(HL_THREADS), lds, stream, L, p); break;
    case 2: hipLaunchKernelGGL(head_gaussian_nll_kernel<2>, dim3(nb), dim3(HL_THREADS), lds, stream, L, p); break;
    case 4: hipLaunchKernelGGL(head_gaussian_nll_kernel<4>, dim3(nb), dim3(HL_THREADS), lds, stream, L, p); break;
    default: hipLaunchKernelGGL(head_gaussian_nll_kernel<8>, dim3(nb), dim3(HL_THREADS), lds, stream, L, p); break;
  }
  GA_CHECK_LAUNCH("head_gaussian_nll");
  hipLaunchKernelGGL(gaussian_nll_finalize_kernel, dim3(1), dim3(64), 0, stream,
                     (const double*)workspace, nb, M, loss_out, grad_slab0,
                     slab_stride, n_splits);
  GA_CHECK_LAUNCH("gaussian_nll_finalize");
  return GA_OK;
}
