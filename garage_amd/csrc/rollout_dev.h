// Device-side pieces of the rollout step shared by rollout.hip and the fused
// policy + env step of policy_fused.hip: Philox, the synthetic environment, the
// NormalizedEnv statistics, the per-step bookkeeping of VecWorker.step_episode
// (sampler/vec_worker.py:176-204).  One thread owns one env.
#pragma once
#include "common.h"

namespace ga_rollout {

// ---- Philox4x32-10 (Random123; Salmon et al. SC'11) --------------------------
struct U4 { uint32_t x, y, z, w; };

static __device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2,
                                            uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}

// The env's arithmetic is numpy's: every product and sum rounded on its own.  The
// __fmul_rn / __fadd_rn intrinsics are plain operators in this toolchain and may
// be contracted into FMAs with their neighbours, so the functions below switch
// contraction off for their own statements instead.
//
// uint32 -> fp32 uniform on [-sqrt3, sqrt3): every step exact or singly rounded,
// so the CPU twin (oracle/envs.py) reproduces it bit for bit.
static __device__ __forceinline__ float u32_unit_variance(uint32_t u) {
#pragma clang fp contract(off)
  const float f = (float)(u >> 8) * 1.1920928955078125e-07f;  // 2^-23
  const float c = f - 1.0f;
  return c * 1.7320508f;
}
static __device__ __forceinline__ float u32_unit_interval(uint32_t u) {
  return ((float)(u >> 8) + 0.5f) * 5.9604644775390625e-08f;  // (0,1)
}

static constexpr uint32_t STREAM_OBS = 0, STREAM_REWARD = 1, STREAM_LENGTH = 2;
static constexpr uint32_t STREAM_ACTION = 3;

// ---- synthetic environment ---------------------------------------------------
struct SynthEnv {
  int64_t n;
  int64_t env_id0;       // global id of env 0 of this shard
  int obs_dim, act_dim, discrete;
  int min_len, max_len;
  uint32_t k0, k1;       // seed
  int32_t* episode;      // [n] episode counter (-1 before the first reset)
  int32_t* t;            // [n] steps taken in the current episode
  int32_t* len;          // [n] length of the current episode
};

static __device__ __forceinline__ void synth_obs(const SynthEnv& e, uint32_t env,
                                          uint32_t episode, uint32_t t, float* out) {
  for (int b = 0; b * 4 < e.obs_dim; ++b) {
    const U4 r = philox4x32_10(env, episode, t, (STREAM_OBS << 16) | (uint32_t)b,
                               e.k0, e.k1);
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
    for (int j = 0; j < 4 && b * 4 + j < e.obs_dim; ++j)
      out[b * 4 + j] = u32_unit_variance(w[j]);
  }
}

static __device__ __forceinline__ int synth_len(const SynthEnv& e, uint32_t env,
                                         uint32_t episode) {
  if (e.min_len >= e.max_len) return e.max_len;
  const U4 r = philox4x32_10(env, episode, 0, STREAM_LENGTH << 16, e.k0, e.k1);
  return e.min_len + (int)(r.x % (uint32_t)(e.max_len - e.min_len + 1));
}

// reset envs where mask != 0 (mask == null: all); writes the first observation.
static __device__ __forceinline__ void synth_reset_one(const SynthEnv& e, int64_t i,
                                                float* obs, int64_t ldo) {
  const uint32_t env = (uint32_t)(e.env_id0 + i);
  const int ep = e.episode[i] + 1;
  e.episode[i] = ep;
  e.t[i] = 0;
  e.len[i] = synth_len(e, env, (uint32_t)ep);
  synth_obs(e, env, (uint32_t)ep, 0u, obs + i * ldo);
}

// What a step of env i reads of the env's and the worker's state: loaded up front,
// so that no load waits behind the step's own stores (the memory counter retires in
// order).  `o` holds the observation entries the reward looks at when they are few.
constexpr int PRE_OBS = 8;
struct EnvPre {
  int ep, t, len, ep_t;
  float o[PRE_OBS];
  bool has_o;
};
static __device__ __forceinline__ int synth_reward_width(const SynthEnv& e) {
  return e.discrete ? e.obs_dim : min(e.act_dim, e.obs_dim);
}

// one env step: reward, step type and the (true) next observation.  `a` is the
// env's action row (any address space), `o` its observation row.
static __device__ __forceinline__ void synth_step_core(
    const SynthEnv& e, int64_t i, const EnvPre& s, const float* a, const float* o,
    float* next_row, float* reward, uint8_t* step_type) {
#pragma clang fp contract(off)
  const uint32_t env = (uint32_t)(e.env_id0 + i);
  const uint32_t ep = (uint32_t)s.ep;
  const int t = s.t;
  const U4 r = philox4x32_10(env, ep, (uint32_t)t, STREAM_REWARD << 16, e.k0, e.k1);
  const float noise = u32_unit_variance(r.x);
  float shaped = 0.f;
  if (e.discrete) {
    const int k = ((int)a[0]) % e.obs_dim;
    if (s.has_o) {
#pragma unroll
      for (int j = 0; j < PRE_OBS; ++j) shaped = j == k ? s.o[j] : shaped;
    } else {
      shaped = o[k];
    }
  } else {
    const int m = min(e.act_dim, e.obs_dim);
    if (s.has_o) {
#pragma unroll
      for (int j = 0; j < PRE_OBS; ++j)
        if (j < m) {
          const float aj = fminf(fmaxf(a[j], -1.f), 1.f);
          const float prod = aj * s.o[j];
          shaped = shaped + prod;
        }
    } else {
      for (int j = 0; j < m; ++j) {
        const float aj = fminf(fmaxf(a[j], -1.f), 1.f);
        const float prod = aj * o[j];
        shaped = shaped + prod;
      }
    }
  }
  const float tenth = 0.1f * shaped;
  *reward = noise + tenth;
  const int tn = t + 1;
  e.t[i] = tn;
  synth_obs(e, env, ep, (uint32_t)tn, next_row);
  // StepType.get_step_type (_dtypes.py:42-68): TIMEOUT wins over done
  uint8_t st;
  if (tn >= e.max_len) st = 3;
  else if (tn >= s.len) st = 2;
  else if (tn == 1) st = 0;
  else st = 1;
  *step_type = st;
}

static __device__ __forceinline__ void synth_step_one(
    const SynthEnv& e, int64_t i, const float* actions, int64_t lda, const float* obs,
    float* next_obs, int64_t ldo, float* reward, uint8_t* step_type) {
  EnvPre s;
  s.ep = e.episode[i];
  s.t = e.t[i];
  s.len = e.len[i];
  s.ep_t = 0;
  s.has_o = false;
  float rew;
  uint8_t st;
  synth_step_core(e, i, s, actions + i * lda, obs + i * ldo, next_obs + i * ldo, &rew,
                  &st);
  reward[i] = rew;
  step_type[i] = st;
}

// ---- NormalizedEnv observation / reward path -----------------------------------
// envs/normalized_env.py:118-132,134-164: per-env exponential moving mean and
// variance (float64 state, alpha = 0.001 by default); the mean is updated first,
// the variance uses the NEW mean, and the value is normalised with the updated
// statistics.  One thread per env; rows with mask == 0 are left untouched.
static __device__ __forceinline__ void obs_normalize_one(const float* src, float* dst,
                                                  double* m, double* v, int obs_dim,
                                                  double alpha) {
#pragma clang fp contract(off)
  for (int j = 0; j < obs_dim; ++j) {
    const double x = (double)src[j];
    const double mn = (1.0 - alpha) * m[j] + alpha * x;
    const double d = x - mn;
    const double vn = (1.0 - alpha) * v[j] + alpha * (d * d);
    m[j] = mn;
    v[j] = vn;
    dst[j] = (float)((x - mn) / (sqrt(vn) + 1e-8));
  }
}

static __device__ __forceinline__ float reward_normalize_one(float reward, double* mean,
                                                      double* var, double alpha,
                                                      double scale, int normalize) {
#pragma clang fp contract(off)
  double r = (double)reward;
  if (normalize) {  // normalized_env.py:126-132,153-164
    const double mn = (1.0 - alpha) * *mean + alpha * r;
    const double d = r - mn;
    const double vn = (1.0 - alpha) * *var + alpha * (d * d);
    *mean = mn;
    *var = vn;
    r = r / (sqrt(vn) + 1e-8);
  }
  return (float)(r * scale);
}

// ---- per-step bookkeeping (VecWorker.step_episode, vec_worker.py:176-204) ------
struct RecordParams {
  int64_t n, col, Tcap;
  int max_episode_length;
  const float* reward;       // [n]
  const uint8_t* step_type;  // [n]
  const float* next_obs;     // [n, ldo]
  int64_t ldo;
  int obs_dim;
  int32_t* ep_t;             // [n] steps so far in the running episode
  float* rew_buf;            // [n, Tcap]
  uint8_t* st_buf;           // [n, Tcap]
  uint16_t* tail_buf;        // [n, Tcap] episode length at its last step, else 0
  float* lastobs_buf;        // [n, Tcap, ldo] written at episode ends only
  uint8_t* done;             // [n] 1 where the env must be reset
  int32_t* step_eps;         // [Tcap] episodes finished at this step
  int32_t* step_samples;     // [Tcap] their total length
  int terminal_only;         // 1: only TERMINAL (not TIMEOUT) ends an episode
};

// bookkeeping of env i; returns the length of the episode that ended (else 0)
static __device__ __forceinline__ int record_core(const RecordParams& p, int64_t i,
                                                  int ep_t, float reward, uint8_t st) {
  int ended_len = 0;
  {
    const int64_t cell = i * p.Tcap + p.col;
    const int t = ep_t + 1;
    // VecWorker ends an episode on any last step (vec_worker.py:198);
    // FragmentWorker only on TERMINAL (fragment_worker.py:114-115)
    const bool ended = (t >= p.max_episode_length) ||
                       (p.terminal_only ? (st == 2) : (st >= 2));
    p.rew_buf[cell] = reward;
    p.st_buf[cell] = st;
    p.tail_buf[cell] = ended ? (uint16_t)t : (uint16_t)0;
    p.done[i] = ended ? 1 : 0;
    p.ep_t[i] = ended ? 0 : t;
    if (ended) {
      ended_len = t;
      const float* o = p.next_obs + i * p.ldo;
      float* lo = p.lastobs_buf + cell * p.ldo;
      for (int j = 0; j < p.obs_dim; ++j) lo[j] = o[j];
    }
  }
  return ended_len;
}

static __device__ __forceinline__ int record_one(const RecordParams& p, int64_t i) {
  return record_core(p, i, p.ep_t[i], p.reward[i], p.step_type[i]);
}

// per-step completion counts: wave-aggregated integer atomics (deterministic:
// integer adds commute)
static __device__ __forceinline__ void record_counts(const RecordParams& p, int ended_len) {
  const uint64_t ballot = __ballot(ended_len > 0);
  int sum = ended_len;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o, 64);
  if ((threadIdx.x & 63) == 0 && ballot) {
    atomicAdd(&p.step_eps[p.col], (int)__popcll(ballot));
    atomicAdd(&p.step_samples[p.col], sum);
  }
}

struct NormParams {
  int norm_obs, norm_reward, scale_reward;
  double* obs_mean;   // [n, obs_dim]
  double* obs_var;
  double obs_alpha;
  double* rew_mean;   // [n]
  double* rew_var;
  double rew_alpha, rew_scale;
};

// env step -> (NormalizedEnv statistics + normalisation) -> bookkeeping -> reset of
// env i when it finished; returns the length of the episode that ended (else 0).
// `raw_obs` / `raw_next` are the env's own observations; seen_next is what the
// policy sees next and what is recorded as the terminal observation -- the same
// buffer as raw_next without normalisation.
struct EnvStepArgs {
  SynthEnv e;
  RecordParams p;
  NormParams nm;
  const float* actions; int64_t lda;
  const float* raw_obs; float* raw_next; float* seen_next;
  float* reward; uint8_t* step_type;
};

static __device__ __forceinline__ EnvPre env_prefetch(const EnvStepArgs& a, int64_t i) {
  EnvPre s;
  s.ep = a.e.episode[i];
  s.t = a.e.t[i];
  s.len = a.e.len[i];
  s.ep_t = a.p.ep_t[i];
  const int m = synth_reward_width(a.e);
  s.has_o = m <= PRE_OBS;
  const float* o = a.raw_obs + i * a.p.ldo;
#pragma unroll
  for (int j = 0; j < PRE_OBS; ++j) s.o[j] = (s.has_o && j < m) ? o[j] : 0.f;
  return s;
}

// `s`: env_prefetch(a, i), taken before anything of this step was stored;
// `act_row`: the env's action (a.actions + i * a.lda, or a copy on chip).
static __device__ __forceinline__ int env_step_one(const EnvStepArgs& a, int64_t i,
                                                   const EnvPre& s,
                                                   const float* act_row) {
  const SynthEnv& e = a.e;
  const RecordParams& p = a.p;
  const NormParams& nm = a.nm;
  float rew;
  uint8_t st;
  synth_step_core(e, i, s, act_row, a.raw_obs + i * p.ldo, a.raw_next + i * p.ldo, &rew,
                  &st);
  a.step_type[i] = st;
  if (nm.norm_obs)  // normalized_env.py:134-151: statistics first, then the value
    obs_normalize_one(a.raw_next + i * p.ldo, a.seen_next + i * p.ldo,
                      nm.obs_mean + i * p.obs_dim, nm.obs_var + i * p.obs_dim, p.obs_dim,
                      nm.obs_alpha);
  if (nm.norm_reward || nm.scale_reward)
    rew = reward_normalize_one(rew, nm.rew_mean + i, nm.rew_var + i, nm.rew_alpha,
                               nm.rew_scale, nm.norm_reward);
  a.reward[i] = rew;
  const int ended_len = record_core(p, i, s.ep_t, rew, st);
  if (ended_len > 0) {
    synth_reset_one(e, i, a.raw_next, p.ldo);
    if (nm.norm_obs)
      obs_normalize_one(a.raw_next + i * p.ldo, a.seen_next + i * p.ldo,
                        nm.obs_mean + i * p.obs_dim, nm.obs_var + i * p.obs_dim,
                        p.obs_dim, nm.obs_alpha);
  }
  return ended_len;
}
static __device__ __forceinline__ int env_step_one(const EnvStepArgs& a, int64_t i) {
  return env_step_one(a, i, env_prefetch(a, i), a.actions + i * a.lda);
}

}  // namespace ga_rollout

// C-ABI argument structs of the rollout step (include/garage_amd.h) and their
// validated conversion (rollout.hip)
struct ga_synth_env;
struct ga_record_args;
struct ga_norm_args;
int ga_build_env_step(const ga_synth_env* env, const ga_record_args* a,
                      const ga_norm_args* norm, const float* actions, int64_t lda,
                      const float* obs, const char* who, ga_rollout::EnvStepArgs* out);
