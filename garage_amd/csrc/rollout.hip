// Rollout-side kernels: action sampling heads, the synthetic batched
// environment, per-step episode bookkeeping and the ragged -> packed compaction.
//
// Together they replace the Python per-env loop of VecWorker.step_episode /
// _gather_episode / collect_episode (sampler/vec_worker.py:139-219) and
// StochasticPolicy.get_actions' dist.sample() (torch/policies/stochastic_policy.py:
// 46-89).  Rollout buffers are env-major (n_envs, Tcap[, width]) in HBM; one
// thread owns one env (its row tails are 16-B friendly: widths are padded to 4).
#include "common.h"

#include "rollout_dev.h"

namespace {
using namespace ga_rollout;

__global__ __launch_bounds__(256) void synth_reset_kernel(SynthEnv e,
                                                          const uint8_t* mask,
                                                          float* obs, int64_t ldo) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= e.n) return;
  if (mask && !mask[i]) return;
  synth_reset_one(e, i, obs, ldo);
}

__global__ __launch_bounds__(256) void synth_step_kernel(
    SynthEnv e, const float* actions, int64_t lda, const float* obs, float* next_obs,
    int64_t ldo, float* reward, uint8_t* step_type) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= e.n) return;
  synth_step_one(e, i, actions, lda, obs, next_obs, ldo, reward, step_type);
}

// src == dst normalises in place; otherwise the raw rows stay untouched (the
// wrapped env keeps its own, un-normalised, state)
__global__ __launch_bounds__(256) void obs_normalize_kernel(
    int64_t n, int obs_dim, const float* src, float* dst, int64_t ldo, double* mean,
    double* var, double alpha, const uint8_t* mask) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (mask && !mask[i]) return;
  obs_normalize_one(src + i * ldo, dst + i * ldo, mean + i * obs_dim,
                    var + i * obs_dim, obs_dim, alpha);
}

__global__ __launch_bounds__(256) void reward_normalize_kernel(
    int64_t n, float* reward, double* mean, double* var, double alpha, double scale,
    int normalize) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  reward[i] = reward_normalize_one(reward[i], normalize ? mean + i : nullptr,
                                   normalize ? var + i : nullptr, alpha, scale,
                                   normalize);
}

// ---- action heads --------------------------------------------------------------
struct HeadParams {
  int64_t n;
  int64_t env_id0;
  int A;                  // action dim (gaussian) or number of classes (categorical)
  const float* head;      // [n, ldh] means or class scores
  int64_t ldh;
  const float* log_std;   // gaussian: device scalar
  int has_min, has_max;
  float min_log_std, max_log_std;
  const float* noise;     // optional [n, ldn]: N(0,1) (gaussian) / U(0,1) (categorical)
  int64_t ldn;
  uint32_t k0, k1;
  uint32_t step;          // global step counter (Philox counter)
  int double_softmax;
  const float* obs;       // [n, ldo] current observations (copied into the buffer)
  int64_t ldo;
  int obs_dim;
  // rollout buffers, column `col`
  int64_t col, Tcap;
  float* action;          // [n, lda] actions handed to the env
  int64_t lda;
  float* obs_buf;         // [n, Tcap, ldo]
  float* act_buf;         // [n, Tcap, lda]
  float* head_buf;        // optional [n, Tcap, ldh]: agent_info 'mean' / probs
};

__device__ __forceinline__ void box_muller(uint32_t u0, uint32_t u1, float* z0,
                                           float* z1) {
  const float a = u32_unit_interval(u0), b = u32_unit_interval(u1);
  const float rad = sqrtf(-2.f * logf(a));
  float s, c;
  sincosf(6.28318530717958647692f * b, &s, &c);
  *z0 = rad * c;
  *z1 = rad * s;
}

// The step's observations into the rollout buffer (the list append of
// vec_worker.py:188), by the whole workgroup: consecutive threads copy consecutive
// columns of a row.  (One thread per env copying its own row -- obs_dim strided
// scalar loads and stores per thread -- cost 156 us per step at C5's 8192 x 376.)
__device__ __forceinline__ void copy_obs_rows(const HeadParams& p) {
  const int64_t env0 = (int64_t)blockIdx.x * 256;
  const int64_t rows = min((int64_t)256, p.n - env0);
  const int64_t total = rows * p.obs_dim;
  for (int64_t e = threadIdx.x; e < total; e += 256) {
    const int64_t env = env0 + e / p.obs_dim;
    const int j = (int)(e % p.obs_dim);
    p.obs_buf[(env * p.Tcap + p.col) * p.ldo + j] = p.obs[env * p.ldo + j];
  }
}

__global__ __launch_bounds__(256) void gaussian_head_kernel(HeadParams p) {
  copy_obs_rows(p);
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.n) return;
  const float s = ga_log_std(*p.log_std, p.has_min, p.min_log_std, p.has_max,
                             p.max_log_std, nullptr);
  const float std = expf(s);
  const uint32_t env = (uint32_t)(p.env_id0 + i);
  const float* mu = p.head + i * p.ldh;
  float* act = p.action + i * p.lda;
  const int64_t cell = i * p.Tcap + p.col;
  float* act_row = p.act_buf + cell * p.lda;
  for (int b = 0; b * 4 < p.A; ++b) {
    float z[4];
    if (p.noise) {
      for (int j = 0; j < 4 && b * 4 + j < p.A; ++j)
        z[j] = p.noise[i * p.ldn + b * 4 + j];
    } else {
      const U4 r = philox4x32_10(env, p.step, (uint32_t)b, STREAM_ACTION << 16, p.k0,
                                 p.k1);
      box_muller(r.x, r.y, &z[0], &z[1]);
      box_muller(r.z, r.w, &z[2], &z[3]);
    }
    for (int j = 0; j < 4 && b * 4 + j < p.A; ++j) {
      const float a = mu[b * 4 + j] + std * z[j];
      act[b * 4 + j] = a;
      act_row[b * 4 + j] = a;
    }
  }
  if (p.head_buf) {
    float* h = p.head_buf + cell * p.ldh;
    for (int j = 0; j < p.A; ++j) h[j] = mu[j];
  }
}

__global__ __launch_bounds__(256) void categorical_head_kernel(HeadParams p) {
  copy_obs_rows(p);
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= p.n) return;
  const float* sc = p.head + i * p.ldh;
  const int64_t cell = i * p.Tcap + p.col;
  // probabilities: softmax(scores), or softmax(softmax(scores))
  float mx = sc[0];
  for (int j = 1; j < p.A; ++j) mx = fmaxf(mx, sc[j]);
  float den = 0.f;
  for (int j = 0; j < p.A; ++j) den += expf(sc[j] - mx);
  float den2 = 0.f;
  if (p.double_softmax)
    for (int j = 0; j < p.A; ++j) den2 += expf(expf(sc[j] - mx) / den);
  float u;
  if (p.noise) {
    u = p.noise[i * p.ldn];
  } else {
    const U4 r = philox4x32_10((uint32_t)(p.env_id0 + i), p.step, 0u,
                               STREAM_ACTION << 16, p.k0, p.k1);
    u = u32_unit_interval(r.x);
  }
  float cdf = 0.f;
  int pick = p.A - 1;
  float* h = p.head_buf ? p.head_buf + cell * p.ldh : nullptr;
  bool found = false;
  for (int j = 0; j < p.A; ++j) {
    float pr = expf(sc[j] - mx) / den;
    if (p.double_softmax) pr = expf(pr) / den2;
    if (h) h[j] = pr;
    cdf += pr;
    if (!found && u < cdf) { pick = j; found = true; }
  }
  p.action[i * p.lda] = (float)pick;
  p.act_buf[cell * p.lda] = (float)pick;
}

__global__ __launch_bounds__(256) void record_step_kernel(RecordParams p) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  record_counts(p, i < p.n ? record_one(p, i) : 0);
}

// env step -> (NormalizedEnv statistics + normalisation) -> bookkeeping -> reset
// of the envs that finished, one thread per env and one launch (every stage only
// touches env i's own state).  `raw_obs` / `raw_next` are the env's own
// observations; p.next_obs is what the policy sees next and what is recorded as
// the terminal observation -- the same buffer as raw_next without normalisation.
__global__ __launch_bounds__(256) void synth_step_record_kernel(EnvStepArgs a) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  record_counts(a.p, i < a.e.n ? env_step_one(a, i) : 0);
}

// ---- ragged -> packed ------------------------------------------------------------
// Episodes in batch order = (completion step, env index).
// One block per completion step t <= t_star ranks the envs that ended there.
__global__ __launch_bounds__(256) void pack_episodes_kernel(
    const uint16_t* tail_buf, int64_t n, int64_t Tcap, const int32_t* ep_base,
    int32_t* ep_env, int32_t* ep_end, int32_t* ep_len) {
  __shared__ int wave_cnt[4];
  __shared__ int running;
  const int t = blockIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x == 0) running = ep_base[t];
  __syncthreads();
  for (int64_t i0 = 0; i0 < n; i0 += 256) {
    const int64_t i = i0 + threadIdx.x;
    const int L = (i < n) ? (int)tail_buf[i * Tcap + t] : 0;
    const uint64_t ballot = __ballot(L > 0);
    const int before = (int)__popcll(ballot & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[w] = (int)__popcll(ballot);
    __syncthreads();
    int base = running;
    for (int k = 0; k < w; ++k) base += wave_cnt[k];
    if (L > 0) {
      const int e = base + before;
      ep_env[e] = (int)i;
      ep_end[e] = t;
      ep_len[e] = L;
    }
    __syncthreads();
    if (threadIdx.x == 0)
      running += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
}

// src[off[e] + j] = env * Tcap + (end - len + 1 + j): flat cell of packed sample.
__global__ __launch_bounds__(256) void pack_src_index_kernel(
    const int32_t* ep_env, const int32_t* ep_end, const int32_t* ep_len,
    const int64_t* ep_off, int64_t n_eps, int64_t Tcap, int32_t* src) {
  const int64_t e = blockIdx.x;
  if (e >= n_eps) return;
  const int L = ep_len[e];
  const int64_t first = (int64_t)ep_env[e] * Tcap + (ep_end[e] - L + 1);
  const int64_t off = ep_off[e];
  for (int j = threadIdx.x; j < L; j += 256) src[off + j] = (int32_t)(first + j);
}

// dst[i, 0:width] = src[idx[i], 0:width]  (width multiple of 4, 16-B vectors)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* src,
                                                          int64_t lds_,
                                                          const int32_t* idx,
                                                          int64_t rows, int width4,
                                                          float* dst, int64_t ldd) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = g / width4;
  const int v = (int)(g % width4);
  if (row >= rows) return;
  const int64_t s = idx[row];
  const float4 x = *reinterpret_cast<const float4*>(src + s * lds_ + 4 * v);
  *reinterpret_cast<float4*>(dst + row * ldd + 4 * v) = x;
}

template <typename T>
__global__ __launch_bounds__(256) void gather_scalar_kernel(const T* src,
                                                            const int32_t* idx,
                                                            int64_t n, T* dst) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}

// per-episode undiscounted reward sums (log_performance, _functions.py:233-275)
__global__ __launch_bounds__(256) void episode_sums_kernel(const float* rewards,
                                                           const int64_t* ep_off,
                                                           int64_t n_eps,
                                                           double* sums) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n_eps) return;
  double acc = 0.0;
  for (int64_t j = ep_off[e]; j < ep_off[e + 1]; ++j) acc += (double)rewards[j];
  sums[e] = acc;
}


// ---- keyed pseudo-random permutation of [0, n) --------------------------------
// BatchDataset (np/optimizers/minibatch_dataset.py:4-35) shuffles ids on the host
// with the global numpy RNG; the throughput mode replaces that by a 4-round
// Feistel network over 2h bits with cycle walking: out[i] = PRP_key(i), computed
// independently per element (no sort, no host round trip).
__device__ __forceinline__ uint32_t feistel_f(uint32_t x, uint32_t k) {
  x ^= k;
  x *= 0x9E3779B1u; x ^= x >> 15;
  x *= 0x85EBCA77u; x ^= x >> 13;
  x *= 0xC2B2AE3Du; x ^= x >> 16;
  return x;
}

__global__ __launch_bounds__(256) void feistel_perm_kernel(int64_t n, int half_bits,
                                                           uint32_t k0, uint32_t k1,
                                                           int32_t* out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t mask = (1u << half_bits) - 1u;
  uint32_t x = (uint32_t)i;
  do {
    uint32_t L = x >> half_bits, R = x & mask;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t t = L ^ (feistel_f(R, k0 + 0x9E3779B9u * r + (k1 ^ r)) & mask);
      L = R;
      R = t;
    }
    x = (L << half_bits) | R;
  } while ((int64_t)x >= n);
  out[i] = (int32_t)x;
}

}  // namespace

// ---------------------------------------------------------------------------
// C ABI (see include/garage_amd.h)
// ---------------------------------------------------------------------------
struct ga_synth_env {
  int64_t n;
  int64_t env_id0;
  int32_t obs_dim, act_dim, discrete, min_len, max_len;
  uint64_t seed;
  int32_t* episode;
  int32_t* t;
  int32_t* len;
};

static SynthEnv to_dev(const ga_synth_env* e) {
  SynthEnv d;
  d.n = e->n; d.env_id0 = e->env_id0; d.obs_dim = e->obs_dim; d.act_dim = e->act_dim;
  d.discrete = e->discrete; d.min_len = e->min_len; d.max_len = e->max_len;
  d.k0 = (uint32_t)(e->seed & 0xffffffffu); d.k1 = (uint32_t)(e->seed >> 32);
  d.episode = e->episode; d.t = e->t; d.len = e->len;
  return d;
}

static int check_env(const ga_synth_env* e, const char* who) {
  GA_REQUIRE(e && e->episode && e->t && e->len, "%s: null env state", who);
  GA_REQUIRE(e->n > 0 && e->obs_dim > 0 && e->act_dim > 0, "%s: bad env sizes", who);
  GA_REQUIRE(e->min_len >= 1 && e->min_len <= e->max_len && e->max_len <= 65535,
             "%s: episode lengths must satisfy 1 <= min <= max <= 65535", who);
  return GA_OK;
}

extern "C" int ga_synth_env_reset(const ga_synth_env* env, const uint8_t* mask,
                                  float* obs, int64_t ldo, hipStream_t stream) {
  int rc = check_env(env, "ga_synth_env_reset");
  if (rc) return rc;
  GA_REQUIRE(obs && ldo >= env->obs_dim, "ga_synth_env_reset: bad obs buffer");
  hipLaunchKernelGGL(synth_reset_kernel, dim3((unsigned)ga_ceil_div(env->n, 256)),
                     dim3(256), 0, stream, to_dev(env), mask, obs, ldo);
  GA_CHECK_LAUNCH("synth_reset");
  return GA_OK;
}

extern "C" int ga_synth_env_step(const ga_synth_env* env, const float* actions,
                                 int64_t lda, const float* obs, float* next_obs,
                                 int64_t ldo, float* reward, uint8_t* step_type,
                                 hipStream_t stream) {
  int rc = check_env(env, "ga_synth_env_step");
  if (rc) return rc;
  GA_REQUIRE(actions && obs && next_obs && reward && step_type,
             "ga_synth_env_step: null pointer");
  GA_REQUIRE(ldo >= env->obs_dim && lda >= (env->discrete ? 1 : env->act_dim),
             "ga_synth_env_step: leading dimensions too small");
  hipLaunchKernelGGL(synth_step_kernel, dim3((unsigned)ga_ceil_div(env->n, 256)),
                     dim3(256), 0, stream, to_dev(env), actions, lda, obs, next_obs,
                     ldo, reward, step_type);
  GA_CHECK_LAUNCH("synth_step");
  return GA_OK;
}

struct ga_head_args {
  int64_t n, env_id0;
  int32_t A, kind;  // kind 0 gaussian, 1 categorical
  const float* head; int64_t ldh;
  const float* log_std; int32_t has_min, has_max; float min_log_std, max_log_std;
  const float* noise; int64_t ldn;
  uint64_t seed; uint32_t step; int32_t double_softmax;
  const float* obs; int64_t ldo; int32_t obs_dim;
  int64_t col, Tcap;
  float* action; int64_t lda;
  float* obs_buf; float* act_buf; float* head_buf;
};

extern "C" int ga_policy_head_sample(const ga_head_args* a, hipStream_t stream) {
  GA_REQUIRE(a && a->head && a->obs && a->action && a->obs_buf && a->act_buf,
             "ga_policy_head_sample: null pointer");
  GA_REQUIRE(a->n > 0 && a->A > 0 && a->ldh >= a->A && a->ldo >= a->obs_dim,
             "ga_policy_head_sample: bad sizes");
  GA_REQUIRE(a->col >= 0 && a->col < a->Tcap, "ga_policy_head_sample: col %lld out of "
             "range (Tcap %lld)", (long long)a->col, (long long)a->Tcap);
  GA_REQUIRE(a->kind == 1 || (a->log_std && a->lda >= a->A),
             "ga_policy_head_sample: gaussian head needs log_std and lda >= A");
  HeadParams p;
  p.n = a->n; p.env_id0 = a->env_id0; p.A = a->A; p.head = a->head; p.ldh = a->ldh;
  p.log_std = a->log_std; p.has_min = a->has_min; p.has_max = a->has_max;
  p.min_log_std = a->min_log_std; p.max_log_std = a->max_log_std; p.noise = a->noise;
  p.ldn = a->ldn; p.k0 = (uint32_t)(a->seed & 0xffffffffu);
  p.k1 = (uint32_t)(a->seed >> 32); p.step = a->step;
  p.double_softmax = a->double_softmax; p.obs = a->obs; p.ldo = a->ldo;
  p.obs_dim = a->obs_dim; p.col = a->col; p.Tcap = a->Tcap; p.action = a->action;
  p.lda = a->lda; p.obs_buf = a->obs_buf; p.act_buf = a->act_buf;
  p.head_buf = a->head_buf;
  const dim3 grid((unsigned)ga_ceil_div(a->n, 256));
  if (a->kind == 0)
    hipLaunchKernelGGL(gaussian_head_kernel, grid, dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL(categorical_head_kernel, grid, dim3(256), 0, stream, p);
  GA_CHECK_LAUNCH("policy_head_sample");
  return GA_OK;
}

struct ga_record_args {
  int64_t n, col, Tcap;
  int32_t max_episode_length;
  const float* reward; const uint8_t* step_type; const float* next_obs;
  int64_t ldo; int32_t obs_dim;
  int32_t* ep_t; float* rew_buf; uint8_t* st_buf; uint16_t* tail_buf;
  float* lastobs_buf; uint8_t* done; int32_t* step_eps; int32_t* step_samples;
  int32_t terminal_only;
};

extern "C" int ga_record_step(const ga_record_args* a, hipStream_t stream) {
  GA_REQUIRE(a && a->reward && a->step_type && a->next_obs && a->ep_t && a->rew_buf &&
                 a->st_buf && a->tail_buf && a->lastobs_buf && a->done &&
                 a->step_eps && a->step_samples,
             "ga_record_step: null pointer");
  GA_REQUIRE(a->n > 0 && a->col >= 0 && a->col < a->Tcap,
             "ga_record_step: col %lld out of range (Tcap %lld)", (long long)a->col,
             (long long)a->Tcap);
  GA_REQUIRE(a->max_episode_length >= 1 && a->max_episode_length <= 65535,
             "ga_record_step: max_episode_length must be in 1..65535");
  RecordParams p;
  p.n = a->n; p.col = a->col; p.Tcap = a->Tcap;
  p.max_episode_length = a->max_episode_length; p.reward = a->reward;
  p.step_type = a->step_type; p.next_obs = a->next_obs; p.ldo = a->ldo;
  p.obs_dim = a->obs_dim; p.ep_t = a->ep_t; p.rew_buf = a->rew_buf;
  p.st_buf = a->st_buf; p.tail_buf = a->tail_buf; p.lastobs_buf = a->lastobs_buf;
  p.done = a->done; p.step_eps = a->step_eps; p.step_samples = a->step_samples;
  p.terminal_only = a->terminal_only;
  hipLaunchKernelGGL(record_step_kernel, dim3((unsigned)ga_ceil_div(a->n, 256)),
                     dim3(256), 0, stream, p);
  GA_CHECK_LAUNCH("record_step");
  return GA_OK;
}

struct ga_norm_args {
  int32_t normalize_obs, normalize_reward;
  double* obs_mean; double* obs_var; double obs_alpha;
  double* reward_mean; double* reward_var; double reward_alpha, reward_scale;
  const float* raw_obs;  // the wrapped env's own current observations
  float* raw_next_obs;   // ... and where its next observations go
};

// Validation + conversion of the C-ABI arguments of one env step (also used by the
// fused policy + env step of policy_fused.hip)
int ga_build_env_step(const ga_synth_env* env, const ga_record_args* a,
                      const ga_norm_args* norm, const float* actions, int64_t lda,
                      const float* obs, const char* who, ga_rollout::EnvStepArgs* out) {
  int rc = check_env(env, who);
  if (rc) return rc;
  GA_REQUIRE(a && a->reward && a->step_type && a->next_obs && a->ep_t && a->rew_buf &&
                 a->st_buf && a->tail_buf && a->lastobs_buf && a->done &&
                 a->step_eps && a->step_samples && actions && obs,
             "%s: null pointer", who);
  GA_REQUIRE(a->n == env->n && a->col >= 0 && a->col < a->Tcap,
             "%s: col %lld out of range (Tcap %lld)", who, (long long)a->col,
             (long long)a->Tcap);
  GA_REQUIRE(a->max_episode_length >= 1 && a->max_episode_length <= 65535,
             "%s: max_episode_length must be in 1..65535", who);
  GA_REQUIRE(a->ldo >= env->obs_dim && a->obs_dim == env->obs_dim &&
                 lda >= (env->discrete ? 1 : env->act_dim),
             "%s: leading dimensions too small", who);
  RecordParams p;
  p.n = a->n; p.col = a->col; p.Tcap = a->Tcap;
  p.max_episode_length = a->max_episode_length; p.reward = a->reward;
  p.step_type = a->step_type; p.next_obs = a->next_obs; p.ldo = a->ldo;
  p.obs_dim = a->obs_dim; p.ep_t = a->ep_t; p.rew_buf = a->rew_buf;
  p.st_buf = a->st_buf; p.tail_buf = a->tail_buf; p.lastobs_buf = a->lastobs_buf;
  p.done = a->done; p.step_eps = a->step_eps; p.step_samples = a->step_samples;
  p.terminal_only = a->terminal_only;
  NormParams nm;
  memset(&nm, 0, sizeof(nm));
  const float* raw_obs = obs;
  float* raw_next = (float*)a->next_obs;
  if (norm) {
    nm.norm_obs = norm->normalize_obs != 0;
    nm.norm_reward = norm->normalize_reward != 0;
    nm.scale_reward = norm->reward_scale != 1.0;
    nm.obs_mean = norm->obs_mean; nm.obs_var = norm->obs_var;
    nm.obs_alpha = norm->obs_alpha; nm.rew_mean = norm->reward_mean;
    nm.rew_var = norm->reward_var; nm.rew_alpha = norm->reward_alpha;
    nm.rew_scale = norm->reward_scale;
    GA_REQUIRE(!nm.norm_obs || (nm.obs_mean && nm.obs_var && norm->raw_obs &&
                                norm->raw_next_obs),
               "%s: observation statistics / raw buffers", who);
    GA_REQUIRE(!nm.norm_reward || (nm.rew_mean && nm.rew_var), "%s: reward statistics",
               who);
    if (nm.norm_obs) {
      raw_obs = norm->raw_obs;
      raw_next = norm->raw_next_obs;
    }
  }
  out->e = to_dev(env); out->p = p; out->nm = nm;
  out->actions = actions; out->lda = lda; out->raw_obs = raw_obs; out->raw_next = raw_next;
  out->seen_next = (float*)a->next_obs; out->reward = (float*)a->reward;
  out->step_type = (uint8_t*)a->step_type;
  return GA_OK;
}

extern "C" int ga_synth_env_step_record_norm(const ga_synth_env* env,
                                             const ga_record_args* a,
                                             const ga_norm_args* norm,
                                             const float* actions, int64_t lda,
                                             const float* obs, hipStream_t stream) {
  EnvStepArgs args;
  int rc = ga_build_env_step(env, a, norm, actions, lda, obs, "ga_synth_env_step_record",
                             &args);
  if (rc) return rc;
  hipLaunchKernelGGL(synth_step_record_kernel, dim3((unsigned)ga_ceil_div(a->n, 256)),
                     dim3(256), 0, stream, args);
  GA_CHECK_LAUNCH("synth_step_record");
  return GA_OK;
}

extern "C" int ga_synth_env_step_record(const ga_synth_env* env,
                                        const ga_record_args* a, const float* actions,
                                        int64_t lda, const float* obs,
                                        hipStream_t stream) {
  return ga_synth_env_step_record_norm(env, a, nullptr, actions, lda, obs, stream);
}

extern "C" int ga_pack_episodes(const uint16_t* tail_buf, int64_t n, int64_t Tcap,
                                int64_t n_steps, const int32_t* ep_base,
                                int32_t* ep_env, int32_t* ep_end, int32_t* ep_len,
                                hipStream_t stream) {
  GA_REQUIRE(tail_buf && ep_base && ep_env && ep_end && ep_len,
             "ga_pack_episodes: null pointer");
  GA_REQUIRE(n > 0 && n_steps > 0 && n_steps <= Tcap, "ga_pack_episodes: bad sizes");
  hipLaunchKernelGGL(pack_episodes_kernel, dim3((unsigned)n_steps), dim3(256), 0,
                     stream, tail_buf, n, Tcap, ep_base, ep_env, ep_end, ep_len);
  GA_CHECK_LAUNCH("pack_episodes");
  return GA_OK;
}

extern "C" int ga_pack_src_index(const int32_t* ep_env, const int32_t* ep_end,
                                 const int32_t* ep_len, const int64_t* ep_off,
                                 int64_t n_eps, int64_t Tcap, int32_t* src,
                                 hipStream_t stream) {
  GA_REQUIRE(ep_env && ep_end && ep_len && ep_off && src,
             "ga_pack_src_index: null pointer");
  GA_REQUIRE(n_eps > 0 && n_eps < (1ll << 31), "ga_pack_src_index: bad n_eps");
  hipLaunchKernelGGL(pack_src_index_kernel, dim3((unsigned)n_eps), dim3(256), 0,
                     stream, ep_env, ep_end, ep_len, ep_off, n_eps, Tcap, src);
  GA_CHECK_LAUNCH("pack_src_index");
  return GA_OK;
}

extern "C" int ga_gather_rows_f32(const float* src, int64_t ld_src,
                                  const int32_t* idx, int64_t rows, int64_t width,
                                  float* dst, int64_t ld_dst, hipStream_t stream) {
  GA_REQUIRE(src && idx && dst, "ga_gather_rows_f32: null pointer");
  GA_REQUIRE(rows > 0 && width > 0 && width % 4 == 0 && ld_src % 4 == 0 &&
                 ld_dst % 4 == 0 && ld_src >= width && ld_dst >= width,
             "ga_gather_rows_f32: widths / strides must be multiples of 4");
  GA_REQUIRE(ga_aligned16(src) && ga_aligned16(dst),
             "ga_gather_rows_f32: 16-B alignment required");
  const int w4 = (int)(width / 4);
  hipLaunchKernelGGL(gather_rows_kernel,
                     dim3((unsigned)ga_ceil_div(rows * w4, 256)), dim3(256), 0, stream,
                     src, ld_src, idx, rows, w4, dst, ld_dst);
  GA_CHECK_LAUNCH("gather_rows");
  return GA_OK;
}

extern "C" int ga_gather_f32(const float* src, const int32_t* idx, int64_t n,
                             float* dst, hipStream_t stream) {
  GA_REQUIRE(src && idx && dst && n > 0, "ga_gather_f32: bad arguments");
  hipLaunchKernelGGL(gather_scalar_kernel<float>,
                     dim3((unsigned)ga_ceil_div(n, 256)), dim3(256), 0, stream, src,
                     idx, n, dst);
  GA_CHECK_LAUNCH("gather_f32");
  return GA_OK;
}

extern "C" int ga_gather_u8(const uint8_t* src, const int32_t* idx, int64_t n,
                            uint8_t* dst, hipStream_t stream) {
  GA_REQUIRE(src && idx && dst && n > 0, "ga_gather_u8: bad arguments");
  hipLaunchKernelGGL(gather_scalar_kernel<uint8_t>,
                     dim3((unsigned)ga_ceil_div(n, 256)), dim3(256), 0, stream, src,
                     idx, n, dst);
  GA_CHECK_LAUNCH("gather_u8");
  return GA_OK;
}

extern "C" int ga_episode_sums_f32(const float* rewards, const int64_t* ep_off,
                                   int64_t n_eps, double* sums, hipStream_t stream) {
  GA_REQUIRE(rewards && ep_off && sums && n_eps > 0, "ga_episode_sums_f32: bad args");
  hipLaunchKernelGGL(episode_sums_kernel, dim3((unsigned)ga_ceil_div(n_eps, 256)),
                     dim3(256), 0, stream, rewards, ep_off, n_eps, sums);
  GA_CHECK_LAUNCH("episode_sums");
  return GA_OK;
}

extern "C" int ga_permutation_i32(int64_t n, uint64_t key, int32_t* out,
                                  hipStream_t stream) {
  GA_REQUIRE(out && n > 0 && n < (1ll << 30), "ga_permutation_i32: bad arguments");
  int half_bits = 1;
  while ((1ll << (2 * half_bits)) < n) ++half_bits;
  hipLaunchKernelGGL(feistel_perm_kernel, dim3((unsigned)ga_ceil_div(n, 256)),
                     dim3(256), 0, stream, n, half_bits,
                     (uint32_t)(key & 0xffffffffu), (uint32_t)(key >> 32), out);
  GA_CHECK_LAUNCH("feistel_perm");
  return GA_OK;
}

extern "C" int ga_obs_normalize_from_f64(int64_t n, int obs_dim, const float* src,
                                         float* dst, int64_t ldo, double* mean,
                                         double* var, double alpha,
                                         const uint8_t* mask, hipStream_t stream) {
  GA_REQUIRE(src && dst && mean && var, "ga_obs_normalize_from_f64: null pointer");
  GA_REQUIRE(n > 0 && obs_dim > 0 && ldo >= obs_dim,
             "ga_obs_normalize_from_f64: bad sizes");
  hipLaunchKernelGGL(obs_normalize_kernel, dim3((unsigned)ga_ceil_div(n, 256)),
                     dim3(256), 0, stream, n, obs_dim, src, dst, ldo, mean, var, alpha,
                     mask);
  GA_CHECK_LAUNCH("obs_normalize");
  return GA_OK;
}

extern "C" int ga_obs_normalize_f64(int64_t n, int obs_dim, float* obs, int64_t ldo,
                                    double* mean, double* var, double alpha,
                                    const uint8_t* mask, hipStream_t stream) {
  return ga_obs_normalize_from_f64(n, obs_dim, obs, obs, ldo, mean, var, alpha, mask,
                                   stream);
}

// NormalizedEnv.step's action rescale (envs/normalized_env.py:90-100): fp32, one
// rounding per numpy operation (no fused multiply-add), np.clip's comparisons (a
// NaN stays a NaN)
__global__ __launch_bounds__(256) void action_rescale_kernel(
    int64_t n, int A, const float* __restrict__ act, int64_t lda,
    const float* __restrict__ lb, const float* __restrict__ ub, float scale,
    float* __restrict__ out, int64_t ldo) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * A) return;
  const int64_t r = i / A;
  const int j = (int)(i % A);
  const float a = act[r * lda + j];
  const float lo = lb[j], hi = ub[j];
  const float slope = (0.5f * (hi - lo)) / scale;
  float v = lo + (a + scale) * slope;
  v = v < lo ? lo : (v > hi ? hi : v);
  out[r * ldo + j] = v;
}

extern "C" int ga_action_rescale_f32(int64_t n, int A, const float* actions, int64_t lda,
                                     const float* low, const float* high,
                                     float expected_action_scale, float* out,
                                     int64_t ldo, hipStream_t stream) {
  GA_REQUIRE(actions && low && high && out, "ga_action_rescale_f32: null pointer");
  GA_REQUIRE(n > 0 && A > 0 && lda >= A && ldo >= A, "ga_action_rescale_f32: bad sizes");
  hipLaunchKernelGGL(action_rescale_kernel, dim3((unsigned)ga_ceil_div(n * A, 256)),
                     dim3(256), 0, stream, n, A, actions, lda, low, high,
                     expected_action_scale, out, ldo);
  GA_CHECK_LAUNCH("action_rescale");
  return GA_OK;
}

extern "C" int ga_reward_normalize_f64(int64_t n, float* reward, double* mean,
                                       double* var, double alpha, double scale,
                                       int normalize, hipStream_t stream) {
  GA_REQUIRE(reward && (!normalize || (mean && var)),
             "ga_reward_normalize_f64: null pointer");
  GA_REQUIRE(n > 0, "ga_reward_normalize_f64: bad size");
  hipLaunchKernelGGL(reward_normalize_kernel, dim3((unsigned)ga_ceil_div(n, 256)),
                     dim3(256), 0, stream, n, reward, mean, var, alpha, scale,
                     normalize);
  GA_CHECK_LAUNCH("reward_normalize");
  return GA_OK;
}
