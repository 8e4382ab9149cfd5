// `n_steps` vectorised rollout steps of the synthetic environment enqueued from
// C++: fused policy step, then env step -> bookkeeping -> reset of finished envs
// in one launch, ping-ponging the two observation buffers.  Same per-env
// operations and order as GpuVecWorker._step drives from Python
// (VecWorker.step_episode, sampler/vec_worker.py:176-204); it exists because the
// Python/ctypes overhead per launch (~12 us) exceeds the device time of a step.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/garage_amd.h"

void ga_set_error(const char* fmt, ...);

// 1 (default): policy step and env step of a rollout step in ONE launch
// (ga_policy_env_step_fused_f32) unless the actions are rescaled in between;
// 0, or GARAGE_AMD_FUSED_ENV_STEP=0 in the environment: two launches
static int g_fused_env_step = -1;
extern "C" int ga_set_fused_env_step(int on) {
  g_fused_env_step = on != 0;
  return 0;
}
static int fused_env_step_on() {
  if (g_fused_env_step < 0) {
    const char* e = getenv("GARAGE_AMD_FUSED_ENV_STEP");
    g_fused_env_step = (e && e[0] == '0') ? 0 : 1;
  }
  return g_fused_env_step;
}

extern "C" int ga_rollout_synth_steps(const ga_mlp_desc* desc, const float* params,
                                      const ga_head_args* head,
                                      const ga_synth_env* env,
                                      const ga_record_args* rec, float* obs_a,
                                      float* obs_b, const ga_norm_args* norm,
                                      float* raw_a, float* raw_b, int64_t n_steps,
                                      ga_stream_t stream) {
  if (!desc || !params || !head || !env || !rec || !obs_a || !obs_b) {
    ga_set_error("ga_rollout_synth_steps: null pointer");
    return -1;
  }
  if (norm && norm->act_low && (!norm->act_high || !norm->scaled_action || env->discrete)) {
    ga_set_error("ga_rollout_synth_steps: action rescale needs bounds, scratch and a "
                 "continuous action space");
    return -1;
  }
  const bool norm_obs = norm && norm->normalize_obs;
  if (norm_obs && (!raw_a || !raw_b)) {
    ga_set_error("ga_rollout_synth_steps: observation normalisation needs the raw "
                 "observation buffers");
    return -1;
  }
  if (n_steps < 0 || head->col + n_steps > head->Tcap) {
    ga_set_error("ga_rollout_synth_steps: steps exceed the rollout buffer");
    return -1;
  }
  if (!ga_policy_step_fused_supported(desc)) {
    ga_set_error("ga_rollout_synth_steps: network not supported by the fused step");
    return -1;
  }
  float* cur = obs_a;
  float* nxt = obs_b;
  float* raw_cur = raw_a;
  float* raw_nxt = raw_b;
  ga_head_args h = *head;
  ga_record_args r = *rec;
  ga_norm_args nm;
  if (norm) nm = *norm;
  if (fused_env_step_on() && !(norm && nm.act_low) && !head->noise && n_steps >= 1) {
    // every rollout step -- policy, env, bookkeeping, reset of the finished envs --
    // in ONE launch for all n_steps: a workgroup owns its envs for the whole rollout
    r.col = h.col;
    r.next_obs = nxt;
    h.obs = cur;
    if (norm) {
      nm.raw_obs = raw_cur;
      nm.raw_next_obs = raw_nxt;
    }
    return ga_policy_env_step_fused_f32(desc, params, &h, env, &r, norm ? &nm : nullptr,
                                        n_steps, stream);
  }
  for (int64_t s = 0; s < n_steps; ++s) {
    h.col = head->col + s;
    h.step = head->step + (uint32_t)s;
    h.obs = cur;
    r.col = h.col;
    r.next_obs = nxt;
    if (norm) {
      nm.raw_obs = raw_cur;
      nm.raw_next_obs = raw_nxt;
    }
    int rc;
    rc = ga_policy_step_fused_f32(desc, params, &h, stream);
    if (rc) return rc;
    // env step -> bookkeeping -> reset of the finished envs: one launch
    const float* env_action = h.action;
    if (norm && nm.act_low) {
      // NormalizedEnv.step: the wrapped env sees the rescaled, clipped action; the
      // batch keeps the policy's own (normalized_env.py:90-114)
      rc = ga_action_rescale_f32(env->n, env->act_dim, h.action, h.lda, nm.act_low,
                                 nm.act_high, nm.expected_action_scale,
                                 nm.scaled_action, h.lda, stream);
      if (rc) return rc;
      env_action = nm.scaled_action;
    }
    rc = ga_synth_env_step_record_norm(env, &r, norm ? &nm : nullptr, env_action, h.lda,
                                       cur, stream);
    if (rc) return rc;
    float* t = cur;
    cur = nxt;
    nxt = t;
    t = raw_cur;
    raw_cur = raw_nxt;
    raw_nxt = t;
  }
  return 0;
}
