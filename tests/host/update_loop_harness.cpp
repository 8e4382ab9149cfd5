// Host-logic harness for the C++ epoch loops (garage_amd/csrc/update.cpp,
// rollout_loop.cpp), built with -fsanitize=address,undefined on the CPU
// (`make asan-host`; SURVEY.md section 5 "sanitizers": GPU sanitizers are not
// available, the host loops are plain C++).  Every kernel entry point the loops
// call -- and the handful of HIP runtime calls they make -- is replaced by a fake
// that records the call, so the checks are about the loops' own arithmetic: which
// rows form which minibatch, how many steps a pass takes (single process and data
// parallel), the order in which two passes interleave, what is skipped in which
// phase, and that argument errors are refused before anything is launched.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/garage_amd.h"
#include "../../garage_amd/csrc/fused_train.h"
#include "../../garage_amd/csrc/small_step.h"

static std::vector<std::string> g_log;
static std::string g_error;

static void logf(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_log.push_back(buf);
}

void ga_set_error(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_error = buf;
}

// ---- HIP runtime fakes -------------------------------------------------------
extern "C" {
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
  logf("wait stream=%p event=%p", (void*)s, (void*)e);
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
  logf("record stream=%p event=%p", (void*)s, (void*)e);
  return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) {
  static char slots[8];
  static int next = 0;
  *e = (hipEvent_t)&slots[next++ % 8];
  return hipSuccess;
}
hipError_t hipMemsetAsync(void*, int, size_t bytes, hipStream_t) {
  logf("memset %zu", bytes);
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }

// ---- kernel entry point fakes --------------------------------------------------
int ga_mlp_forward_f32(const ga_mlp_desc* d, const float*, const float*, int64_t,
                       const int32_t* idx, int64_t M, float*, float* out, int64_t,
                       ga_stream_t s) {
  logf("fwd M=%lld idx0=%d layers=%d out=%d stream=%p", (long long)M, idx ? idx[0] : -1,
       d->n_layers, out != nullptr, s);
  if (idx) {  // touch every id of the minibatch: out-of-range slices trip ASAN
    long long sum = 0;
    for (int64_t i = 0; i < M; ++i) sum += idx[i];
    logf("idxsum %lld", sum);
  }
  return 0;
}
int ga_ppo_gaussian_loss_f32(const float*, int64_t, const float*, int64_t, const float*,
                             const float*, const int32_t*, const float*, int, float, int,
                             float, int64_t M, int A, int algo, float, float, int, float*,
                             float*, float* loss_out, float*, int64_t, int64_t splits,
                             double*, ga_stream_t) {
  logf("ppo_loss M=%lld A=%d algo=%d splits=%lld", (long long)M, A, algo, (long long)splits);
  if (loss_out) *loss_out = (float)M;
  return 0;
}
int ga_ppo_categorical_loss_f32(const float*, int64_t, const float*, int64_t, const float*,
                                const float*, const int32_t*, int64_t M, int, int, int,
                                float, float, int, float*, float*, float*, float*, double*,
                                float*, int64_t, int64_t, double*, ga_stream_t) {
  logf("cat_loss M=%lld", (long long)M);
  return 0;
}
int ga_gaussian_nll_loss_f32(const float*, int64_t, const float*, const int32_t*,
                             const float*, int64_t M, float*, float*, float*, int64_t,
                             int64_t, double*, ga_stream_t) {
  logf("nll_loss M=%lld", (long long)M);
  return 0;
}
int ga_head_loss_supported(int, int) { return 0; }
int ga_head_ppo_gaussian_loss_f32(const float*, int64_t, const float*, int64_t,
                                  const float*, int, float*, int64_t, const float*, int64_t,
                                  const float*, const float*, const int32_t*, const float*,
                                  int, float, int, float, int64_t, int, int, float, float,
                                  int, float*, int64_t, float*, float*, float*, int64_t,
                                  int64_t, double*, ga_stream_t) {
  return 0;
}
int ga_head_gaussian_nll_loss_f32(const float*, int64_t, const float*, const float*, int,
                                  float*, int64_t, const float*, const int32_t*,
                                  const float*, int64_t, float*, int64_t, float*, float*,
                                  int64_t, int64_t, double*, ga_stream_t) {
  return 0;
}
int64_t ga_mlp_backward_splits(const ga_mlp_desc*, int64_t M) { return (M + 255) / 256; }
int ga_mlp_backward_f32(const ga_mlp_desc*, const float*, const float*, int64_t,
                        const int32_t*, int64_t M, const float*, const float*, int64_t,
                        float*, float*, int64_t, int64_t splits, ga_stream_t) {
  logf("bwd M=%lld splits=%lld", (long long)M, (long long)splits);
  return 0;
}
int ga_mlp_backward_range_f32(const ga_mlp_desc*, const float*, const float*, int64_t,
                              const int32_t*, int64_t M, const float*, const float*,
                              int64_t, float*, float*, int64_t, int64_t, int l_start,
                              int fused_first, ga_stream_t) {
  logf("bwd_range M=%lld l_start=%d fused_first=%d", (long long)M, l_start, fused_first);
  return 0;
}
int ga_reduce_adam_f32(const float*, int64_t splits, int64_t, float*, float*, float*,
                       float*, int64_t, int64_t step, double, double, double, double,
                       int zero0, ga_stream_t) {
  logf("reduce_adam splits=%lld step=%lld zero0=%d", (long long)splits, (long long)step,
       zero0);
  return 0;
}
int ga_reduce_slabs_f32(const float*, int64_t splits, int64_t, int64_t, float scale,
                        float*, ga_stream_t) {
  logf("reduce_slabs splits=%lld scale=%.4f", (long long)splits, scale);
  return 0;
}
int ga_adam_step_f32(float*, const float*, float*, float*, int64_t, int64_t step, double,
                     double, double, double, ga_stream_t) {
  logf("adam step=%lld", (long long)step);
  return 0;
}
int64_t ga_reduction_partials_doubles(void) { return 1024; }
int ga_small_step_supported(int n_layers, const int* dims, int64_t M) {
  return n_layers == 3 && dims[1] == dims[2] && dims[1] % 32 == 0 && dims[1] <= 256 &&
         dims[0] <= 32 && dims[3] <= 8 && M >= 1 && M <= 64;
}
int ga_small_step_resident(int, int) { return 1; }
// (check 9 sets these: floats behind a->params / a->xh2 / a->xdz that the caller owns)
static int64_t g_ss_n_flat = 0, g_ss_ws_floats = 0;
static int g_ss_extent_errors = 0;
int ga_small_step(const ga_small_step_args* a, void*) {
  logf("small_step M=%d step=%lld", a->M, (long long)a->step);
  if (g_ss_n_flat > 0) {
    // the largest offsets small_step_kernel's index formulas reach (small_step.hip),
    // last workgroup (c0 = H - 16), idle waves included -- their prefetches are
    // clamped to the layer's last tile, which is what this arithmetic restates
    const int64_t H = a->H, ld0 = (a->in_w + 3) & ~3, A = a->out_w;
    const int64_t c0 = H - 16;
    int64_t reach[8];
    reach[0] = a->w_off[0] + H * ld0 - 1;                              // W1 staging
    reach[1] = a->w_off[1] + (H - 1) * H + c0 + 15;                    // W2 columns
    reach[2] = a->w_off[1] + (c0 + 15) * H + H - 1;                    // W2 own rows
    reach[3] = a->w_off[1] + (c0 + 15) * H + 32 * (H / 32 - 1) + 31;   // Adam prefetch
    reach[4] = a->w_off[2] + (A - 1) * H + c0 + 15;                    // head columns
    reach[5] = a->b_off[0] + H - 1;
    reach[6] = a->b_off[1] + c0 + 15;
    reach[7] = a->b_off[2] + A - 1;
    for (int i = 0; i < 8; ++i)
      if (reach[i] >= g_ss_n_flat) ++g_ss_extent_errors;
    // the two exchanges: head shares [H / 16][64][8], dZ2 [64][H] -- written here
    // so that AddressSanitizer sees the extents
    memset(a->xh2, 0, sizeof(float) * (size_t)((H / 16) * 64 * 8));
    memset(a->xdz, 0, sizeof(float) * (size_t)(64 * H));
    if ((H / 16) * 64 * 8 > g_ss_ws_floats || 64 * H > g_ss_ws_floats)
      ++g_ss_extent_errors;
    // and the optimizer state at the largest index
    a->params[reach[3]] = a->exp_avg[reach[3]] = a->exp_avg_sq[reach[3]] = 0.f;
  }
  return 0;
}
int ga_act_slope_mul_f32(float*, int64_t, const float*, int64_t, int64_t M, int N, int act,
                         void*) {
  logf("act_slope_mul M=%lld N=%d act=%d", (long long)M, N, act);
  return 0;
}
int ga_fused_width_ok(int w) { return w == 64 || w == 128 || w == 256; }
int ga_fused_first_layer_ok(int in_w, int K) {
  return in_w >= 1 && in_w <= 32 && K % 32 == 0 && K * ((in_w + 3) & ~3) <= 5120;
}
int64_t ga_fused_tiles(int64_t M) { return (M + 63) / 64; }
int ga_fused_fwd_head_loss(const float*, int64_t, const int32_t* a_idx, const float*,
                           int64_t, const float*, int64_t M, int width, int K,
                           const float*, int64_t, const float*,
                           const ga_fused_loss_args* loss, float*, int64_t, float* hpart,
                           double* lpart, const ga_fused_first_layer* first,
                           hipStream_t) {
  logf("fused_fwd M=%lld width=%d K=%d a_idx=%d A=%d first=%d", (long long)M, width, K,
       a_idx != nullptr, loss->A, first != nullptr);
  // the partial-sum scratch must hold what the plan says
  const int64_t tiles = ga_fused_tiles(M);
  for (int64_t t = 0; t < tiles; ++t) {
    lpart[2 * t] = 0.0;
    lpart[2 * t + 1] = 0.0;
    memset(hpart + t * (8 * (int64_t)width + 8), 0, sizeof(float) * (8 * width + 8));
  }
  return 0;
}
int ga_fused_dgrad_wgrad0(const float*, int64_t, const float*, int64_t, int64_t M,
                          int width, int K, const float*, int64_t, const float*, int64_t,
                          const int32_t*, int in_w, float* wpart, hipStream_t) {
  logf("fused_dgrad M=%lld width=%d K=%d in=%d", (long long)M, width, K, in_w);
  const int64_t ld0 = (in_w + 3) & ~3;
  const int64_t tiles = ga_fused_tiles(M);
  for (int64_t t = 0; t < tiles; ++t)
    memset(wpart + t * (width * ld0 + width), 0, sizeof(float) * (width * ld0 + width));
  return 0;
}
int ga_reduce_regions_adam(const ga_fused_region* r, int n, float*, float*, float*, float*,
                           int64_t step, double, double, double, double, float scale,
                           int do_adam, int zero0, const double*, int n_lpart, int64_t M,
                           const ga_fused_loss_args*, float* loss_out, hipStream_t) {
  logf("reduce_regions n=%d step=%lld scale=%.4f adam=%d zero0=%d lparts=%d M=%lld", n,
       (long long)step, scale, do_adam, zero0, n_lpart, (long long)M);
  for (int k = 0; k < n; ++k)
    logf("  region beg=%lld n=%lld parts=%d stride=%lld", (long long)r[k].beg,
         (long long)r[k].n, r[k].n_part, (long long)r[k].stride);
  if (loss_out) *loss_out = 1.f;
  return 0;
}
// ---- pair launches (two networks per grid): the fakes write the same extents
int ga_split_bf16_any(void) { return 0; }
void ga_reduce_planes_hint(int64_t, int, int) {}
void ga_planes_epoch_begin(void) {}
int ga_fused_pair_supported(int width, int K, int in_w) {
  return width == 256 && K <= 256 && ga_fused_first_layer_ok(in_w, K);
}
int ga_fused_fwd_head_loss_pair(
    int64_t M, int width, int K, const float*, int64_t, const float*, const float*, int64_t,
    const float*, const ga_fused_loss_args* la, float*, int64_t, float* hpa, double* lpa,
    const ga_fused_first_layer* fa, const float*, int64_t, const float*, const float*,
    int64_t, const float*, const ga_fused_loss_args* lb, float*, int64_t, float* hpb,
    double* lpb, const ga_fused_first_layer* fb, hipStream_t s) {
  logf("pair_fwd stream=%p M=%lld width=%d K=%d A=%d/%d in=%d/%d", (void*)s, (long long)M,
       width, K, la->A, lb->A, fa->in_w, fb->in_w);
  const int64_t tiles = ga_fused_tiles(M);
  float* hp[2] = {hpa, hpb};
  double* lp[2] = {lpa, lpb};
  for (int i = 0; i < 2; ++i)
    for (int64_t t = 0; t < tiles; ++t) {
      lp[i][2 * t] = lp[i][2 * t + 1] = 0.0;
      memset(hp[i] + t * (8 * (int64_t)width + 8), 0, sizeof(float) * (8 * width + 8));
    }
  return 0;
}
int ga_wgrad_mid_pair(int64_t M, int64_t n_splits, int out_w, int in_w, const float*,
                      const float*, float* swa, float* sba, int64_t ssa, const float*,
                      const float*, float* swb, float* sbb, int64_t ssb, hipStream_t s) {
  logf("pair_wgrad stream=%p M=%lld splits=%lld out=%d in=%d", (void*)s, (long long)M,
       (long long)n_splits, out_w, in_w);
  float* sw[2] = {swa, swb};
  float* sb[2] = {sba, sbb};
  const int64_t ss[2] = {ssa, ssb};
  for (int i = 0; i < 2; ++i)
    for (int64_t k = 0; k < n_splits; ++k) {
      memset(sw[i] + k * ss[i], 0, sizeof(float) * (size_t)out_w * in_w);
      memset(sb[i] + k * ss[i], 0, sizeof(float) * (size_t)out_w);
    }
  return 0;
}
int ga_fused_dgrad_wgrad0_pair(int64_t M, int width, int K, int in_w, const float*, int64_t,
                               const float*, int64_t, const float*, int64_t, const float*,
                               int64_t, const int32_t*, float* wpa, const float*, int64_t,
                               const float*, int64_t, const float*, int64_t, const float*,
                               int64_t, const int32_t*, float* wpb, hipStream_t s) {
  logf("pair_dgrad stream=%p M=%lld width=%d K=%d in=%d", (void*)s, (long long)M, width, K,
       in_w);
  const int64_t ld0 = (in_w + 3) & ~3, tiles = ga_fused_tiles(M);
  float* wp[2] = {wpa, wpb};
  for (int i = 0; i < 2; ++i)
    for (int64_t t = 0; t < tiles; ++t)
      memset(wp[i] + t * (width * ld0 + width), 0, sizeof(float) * (width * ld0 + width));
  return 0;
}
int ga_reduce_regions_adam_pair(const ga_reduce_net* a, const ga_reduce_net* b,
                                hipStream_t s) {
  logf("pair_reduce stream=%p steps=%lld/%lld regions=%d/%d adam=%d/%d M=%lld", (void*)s,
       (long long)a->step, (long long)b->step, a->n_regions, b->n_regions, a->do_adam,
       b->do_adam, (long long)a->M);
  if (a->loss_out) *a->loss_out = 1.f;
  if (b->loss_out) *b->loss_out = 1.f;
  return 0;
}
int ga_narrow_step_supported(int n_layers, const int* dims) {
  return n_layers == 3 && dims[1] == dims[2] && (dims[1] == 32 || dims[1] == 64) &&
         dims[0] <= 32 && dims[3] <= 8;
}
int64_t ga_narrow_step_stride(int in_w, int H) {
  const int64_t ld0 = (in_w + 3) & ~3;
  return (int64_t)H * ld0 + H + (int64_t)H * H + H + 8 * (int64_t)H + 8;
}
int ga_narrow_train_step(const float*, const int64_t*, const int64_t*, int in_w, int H,
                         int out_w, const float*, int64_t, int64_t M,
                         const ga_fused_loss_args*, float* part, double* lpart,
                         hipStream_t) {
  logf("narrow M=%lld in=%d H=%d out=%d", (long long)M, in_w, H, out_w);
  const int64_t tiles = ga_fused_tiles(M), stride = ga_narrow_step_stride(in_w, H);
  for (int64_t t = 0; t < tiles; ++t) {
    lpart[2 * t] = lpart[2 * t + 1] = 0.0;
    memset(part + t * stride, 0, sizeof(float) * stride);
  }
  return 0;
}
int ga_policy_step_fused_supported(const ga_mlp_desc*) { return 1; }
int ga_policy_step_fused_f32(const ga_mlp_desc*, const float*, const ga_head_args* h,
                             ga_stream_t) {
  logf("policy_step col=%lld step=%u obs=%p", (long long)h->col, h->step, (void*)h->obs);
  return 0;
}
int ga_policy_env_step_fused_f32(const ga_mlp_desc*, const float*, const ga_head_args* h,
                                 const ga_synth_env*, const ga_record_args* r,
                                 const ga_norm_args* nm, int64_t n_steps, ga_stream_t) {
  logf("policy_env_step col=%lld step=%u obs=%p next=%p norm=%d steps=%lld",
       (long long)h->col, h->step, (void*)h->obs, (void*)r->next_obs, nm != nullptr,
       (long long)n_steps);
  return 0;
}
int ga_synth_env_step_record_norm(const ga_synth_env*, const ga_record_args* r,
                                  const ga_norm_args* nm, const float* act, int64_t,
                                  const float* obs, ga_stream_t) {
  logf("env_step col=%lld obs=%p next=%p act=%p norm=%d", (long long)r->col, (void*)obs,
       (void*)r->next_obs, (void*)act, nm != nullptr);
  return 0;
}
int ga_action_rescale_f32(int64_t n, int A, const float*, int64_t, const float*,
                          const float*, float s, float*, int64_t, ga_stream_t) {
  logf("rescale n=%lld A=%d s=%.2f", (long long)n, A, s);
  return 0;
}
}  // extern "C"

// ---- checks ------------------------------------------------------------------
static int g_failed = 0;
#define CHECK(cond)                                                       \
  do {                                                                    \
    if (!(cond)) {                                                        \
      fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);   \
      ++g_failed;                                                         \
    }                                                                     \
  } while (0)

static int count(const char* prefix) {
  int n = 0;
  for (auto& l : g_log) n += l.rfind(prefix, 0) == 0;
  return n;
}
static std::vector<long long> ms_of(const char* prefix) {
  std::vector<long long> out;
  for (auto& l : g_log)
    if (l.rfind(prefix, 0) == 0) {
      long long m = -1;
      sscanf(l.c_str() + strlen(prefix), " M=%lld", &m);
      out.push_back(m);
    }
  return out;
}

struct Net {
  ga_mlp_desc d;
  std::vector<float> params, acts, slabs, scratch;
  explicit Net(int in, int h1, int h2, int out) {
    memset(&d, 0, sizeof(d));
    d.n_layers = 3;
    d.dims[0] = in; d.dims[1] = h1; d.dims[2] = h2; d.dims[3] = out;
    int64_t off = 4;
    const int w[4] = {in, h1, h2, out};
    for (int l = 0; l < 3; ++l) {
      d.w_off[l] = off; off += (int64_t)w[l + 1] * ((w[l] + 3) & ~3);
      d.b_off[l] = off; off += (w[l + 1] + 3) & ~3;
    }
    d.act_off[0] = 0; d.act_off[1] = 1 << 16;
    params.assign(off, 0.f);
    acts.assign(1 << 18, 0.f);
    slabs.assign(off * 8, 0.f);
    scratch.assign(16, 0.f);
  }
  ga_update_args args(int64_t S, int64_t mb, const int32_t* perm, int kind) {
    ga_update_args a;
    memset(&a, 0, sizeof(a));
    a.desc = &d; a.params = a.grads = a.exp_avg = a.exp_avg_sq = params.data();
    a.n_flat = (int64_t)params.size();
    a.acts = a.dacts = acts.data(); a.out = a.dout = acts.data(); a.ldo = 8;
    a.slabs = slabs.data(); a.max_splits = 8;
    a.lr = 1e-3; a.beta1 = 0.9; a.beta2 = 0.999; a.eps = 1e-8; a.learn_std = 1;
    a.X = params.data(); a.ldx = (d.dims[0] + 3) & ~3; a.S = S; a.perm = perm; a.mb = mb;
    a.kind = kind; a.actions = params.data(); a.lda = 4; a.old_ll = a.adv = a.returns =
        params.data();
    a.loss_scratch = scratch.data();
    static std::vector<double> ws(2048, 0.0);
    a.workspace = ws.data();
    return a;
  }
};

extern "C" int ga_set_small_step(int on);
extern "C" int ga_set_fused_train(int on);
extern "C" int64_t ga_update_partials_floats(const ga_mlp_desc* d, int64_t M);
extern "C" void ga_set_allreduce_hook(ga_allreduce_fn fn);
static int fake_allreduce(void*, float*, int64_t n, void*) {
  logf("allreduce n=%lld", (long long)n);
  return 0;
}

int main() {
  std::vector<int32_t> perm(23);
  for (int i = 0; i < 23; ++i) perm[i] = 22 - i;
  Net net(17, 48, 48, 6);  // 48-wide: neither the small step nor the fused step
  // 1. BatchDataset minibatches: ceil(S / mb) of mb ids, the last one partial
  {
    g_log.clear();
    ga_update_args a = net.args(23, 5, perm.data(), 0);
    CHECK(ga_update_epoch(&a, nullptr) == 0);
    CHECK((ms_of("fwd") == std::vector<long long>{5, 5, 5, 5, 3}));
    CHECK(count("reduce_adam") == 5 && count("adam step") == 0);
    CHECK(g_log[0].find("idx0=22") != std::string::npos);
  }
  // 2. the even split of a data-parallel rank: exactly n_mb steps, per-step scales,
  //    one all-reduce each, Adam after it
  {
    g_log.clear();
    ga_set_allreduce_hook(fake_allreduce);
    const float scales[4] = {0.25f, 0.5f, 0.75f, 1.0f};
    ga_update_args a = net.args(23, 0, perm.data(), 1);
    a.n_mb = 4; a.grad_scales_host = scales; a.comm = (void*)1; a.world = 2;
    a.step0 = 10;
    CHECK(ga_update_epoch(&a, nullptr) == 0);
    CHECK((ms_of("fwd") == std::vector<long long>{5, 6, 6, 6}));
    CHECK(count("allreduce") == 4 && count("reduce_adam") == 0);
    CHECK(count("reduce_slabs splits=1 scale=0.2500") == 1);
    CHECK(count("reduce_slabs splits=1 scale=1.0000") == 1);
    CHECK(count("adam step=11") == 1 && count("adam step=14") == 1);
  }
  // 3. phase 1 stops at the scaled gradient: no exchange, no optimizer
  {
    g_log.clear();
    ga_update_args a = net.args(23, 23, perm.data(), 0);
    a.phase = 1; a.grad_scale = 0.5f;
    CHECK(ga_update_epoch(&a, nullptr) == 0);
    CHECK(count("reduce_slabs splits=1 scale=0.5000") == 1);
    CHECK(count("adam") == 0 && count("allreduce") == 0 && count("reduce_adam") == 0);
  }
  // 4. two passes interleave minibatch by minibatch on their streams
  {
    g_log.clear();
    Net other(17, 48, 48, 1);
    ga_update_args a = net.args(23, 8, perm.data(), 0);
    ga_update_args b = other.args(23, 8, perm.data(), 1);
    std::vector<double> ws2(2048, 0.0);
    b.workspace = ws2.data();
    CHECK(ga_update_epoch_pair(&a, (void*)0x10, &b, (void*)0x20) == 0);
    std::vector<std::string> order;
    for (auto& l : g_log)
      if (l.rfind("fwd", 0) == 0)
        order.push_back(l.find("stream=0x10") != std::string::npos ? "a" : "b");
    CHECK((order == std::vector<std::string>{"a", "b", "a", "b", "a", "b"}));
    // sharing buffers is refused
    CHECK(ga_update_epoch_pair(&a, (void*)0x10, &a, (void*)0x20) != 0);
  }
  // 5. argument errors never launch
  {
    g_log.clear();
    ga_update_args a = net.args(0, 5, perm.data(), 0);
    CHECK(ga_update_epoch(&a, nullptr) != 0 && g_log.empty());
    a = net.args(23, 0, perm.data(), 0);
    CHECK(ga_update_epoch(&a, nullptr) != 0 && g_log.empty());
    a = net.args(23, 5, perm.data(), 0);
    a.n_mb = 24;  // more minibatches than samples
    CHECK(ga_update_epoch(&a, nullptr) != 0 && g_log.empty());
    CHECK(ga_update_epoch(nullptr, nullptr) != 0);
    a = net.args(4000, 4000, nullptr, 0);
    a.max_splits = 2;  // slab workspace too small for 16 splits
    CHECK(ga_update_epoch(&a, nullptr) != 0);
    CHECK(g_error.find("slab workspace") != std::string::npos);
  }
  // 6. the small step takes minibatches of <= 64 rows of a 2 x H net
  {
    g_log.clear();
    Net small(17, 64, 64, 6);
    std::vector<int32_t> p2(200);
    for (int i = 0; i < 200; ++i) p2[i] = i;
    ga_update_args a = small.args(200, 64, p2.data(), 0);
    ga_set_fused_train(0);
    CHECK(ga_update_epoch(&a, nullptr) == 0);
    CHECK(count("small_step") == 4 && count("fwd") == 0);  // 64, 64, 64, 8
    ga_set_small_step(0);
    g_log.clear();
    CHECK(ga_update_epoch(&a, nullptr) == 0);
    CHECK(count("small_step") == 0 && count("fwd") == 4);
    ga_set_small_step(1);
    ga_set_fused_train(1);
  }
  // 7. the fused step: plan, scratch layout, regions
  {
    g_log.clear();
    Net wide(17, 256, 256, 6);
    const int64_t M = 1000;
    const int64_t need = ga_update_partials_floats(&wide.d, M);
    const int64_t tiles = (M + 63) / 64;
    CHECK(need == 4 * tiles + tiles * (8 * 256 + 8) + tiles * (256 * 20 + 256));
    CHECK(ga_update_partials_floats(&net.d, M) == 0);  // 48-wide: per-layer path
    std::vector<float> partials((size_t)need, 1.f);  // exactly what was asked for
    std::vector<int32_t> p3(M);
    for (int i = 0; i < M; ++i) p3[i] = i;
    ga_update_args a = wide.args(M, M, p3.data(), 0);
    a.partials = partials.data(); a.partials_floats = need;
    CHECK(ga_update_epoch(&a, nullptr) == 0);
    // (the first layer is computed inside the fused kernel: no forward launch)
    CHECK(count("fused_fwd M=1000 width=256 K=256 a_idx=0 A=6 first=1") == 1);
    CHECK(count("bwd_range M=1000 l_start=1 fused_first=1") == 1);
    CHECK(count("fused_dgrad M=1000 width=256 K=256 in=17") == 1);
    CHECK(count("reduce_regions n=6 step=1 scale=1.0000 adam=1 zero0=0 lparts=16") == 1);
    CHECK(count("fwd M=1000") == 0);
    ga_set_fused_first_layer(0);
    g_log.clear();
    CHECK(ga_update_epoch(&a, nullptr) == 0);
    CHECK(count("fused_fwd M=1000 width=256 K=256 a_idx=0 A=6 first=0") == 1);
    CHECK(count("fwd M=1000") == 1);  // the hidden layer below the last one
    ga_set_fused_first_layer(1);
    // a scratch one float short falls back to the per-layer kernels
    g_log.clear();
    a.partials_floats = need - 1;
    CHECK(ga_update_epoch(&a, nullptr) == 0);
    CHECK(count("fused_fwd") == 0 && count("reduce_adam") == 1);
  }
  // 7b. 2 x 64 networks: the whole step in one launch + the reduction
  {
    g_log.clear();
    Net narrow(4, 64, 64, 2);
    const int64_t M = 200, tiles = 4;
    const int64_t need = ga_update_partials_floats(&narrow.d, M);
    CHECK(need == 4 * tiles + tiles * ga_narrow_step_stride(4, 64));
    std::vector<float> partials((size_t)need, 1.f);
    std::vector<int32_t> p4(M);
    for (int i = 0; i < M; ++i) p4[i] = i;
    ga_update_args a = narrow.args(M, 100, p4.data(), 2);
    a.partials = partials.data(); a.partials_floats = need;
    CHECK(ga_update_epoch(&a, nullptr) == 0);
    CHECK(count("narrow M=100 in=4 H=64 out=2") == 2 && count("fwd") == 0);
    CHECK(count("reduce_regions n=6 step=1") == 1 && count("reduce_regions n=6 step=2") == 1);
    CHECK(count("  region beg=4 n=256 parts=2") == 2);  // W1, both steps
  }
  // 8. the native rollout loop ping-pongs the observation buffers
  {
    g_log.clear();
    ga_head_args h;
    memset(&h, 0, sizeof(h));
    h.col = 3; h.Tcap = 16; h.step = 100;
    ga_synth_env env;
    memset(&env, 0, sizeof(env));
    env.n = 4; env.act_dim = 2;
    ga_record_args rec;
    memset(&rec, 0, sizeof(rec));
    float A[4], B[4];
    Net pol(17, 64, 64, 6);
    CHECK(ga_rollout_synth_steps(&pol.d, pol.params.data(), &h, &env, &rec, A, B, nullptr,
                                 nullptr, nullptr, 3, nullptr) == 0);
    // (ONE launch for all the steps by default)
    CHECK(count("policy_env_step") == 1 && count("policy_step ") == 0);
    char want[160];
    snprintf(want, sizeof(want),
             "policy_env_step col=3 step=100 obs=%p next=%p norm=0 steps=3", (void*)A,
             (void*)B);
    CHECK(count(want) == 1);
    ga_set_fused_env_step(0);
    g_log.clear();
    CHECK(ga_rollout_synth_steps(&pol.d, pol.params.data(), &h, &env, &rec, A, B, nullptr,
                                 nullptr, nullptr, 3, nullptr) == 0);
    CHECK(count("policy_step") == 3 && count("env_step col") == 3);
    snprintf(want, sizeof(want), "env_step col=4 obs=%p next=%p", (void*)B, (void*)A);
    CHECK(count(want) == 1);
    ga_set_fused_env_step(1);
    CHECK(ga_rollout_synth_steps(&pol.d, pol.params.data(), &h, &env, &rec, A, B, nullptr,
                                 nullptr, nullptr, 14, nullptr) != 0);  // past Tcap
  }
  // 8b. Two 256-wide passes with equal shapes: ga_update_epoch_pair runs step k of
  //     both as four pair launches on stream_a, ordered against stream_b on both
  //     sides; a different minibatch count, a data-parallel communicator or
  //     ga_set_merged_pair(0) give the two-stream schedule back
  {
    extern int ga_set_merged_pair(int on);
    ga_set_fused_train(1);
    ga_set_merged_pair(1);  // (opt-in: the two-stream schedule is the default)
    Net pol(17, 256, 256, 6), vf(17, 256, 256, 1);
    const int64_t S = 1000, mb = 300;  // 4 minibatches: 300, 300, 300, 100
    std::vector<int32_t> pp((size_t)S), pv((size_t)S);
    for (int64_t i = 0; i < S; ++i) { pp[(size_t)i] = (int32_t)i; pv[(size_t)i] = (int32_t)(S - 1 - i); }
    const int64_t need = ga_update_partials_floats(&pol.d, mb);
    CHECK(need > 0 && need == ga_update_partials_floats(&vf.d, mb));
    std::vector<float> partp((size_t)need), partv((size_t)need);
    const int64_t splits = ga_mlp_backward_splits(&pol.d, mb);
    std::vector<float> slp((size_t)(splits * (int64_t)pol.params.size())),
        slv((size_t)(splits * (int64_t)vf.params.size()));
    std::vector<float> ap((size_t)(mb * 512)), dp((size_t)(mb * 512)), av((size_t)(mb * 512)),
        dv((size_t)(mb * 512));
    auto mk = [&](Net& n, std::vector<int32_t>& perm, int kind, std::vector<float>& part,
                  std::vector<float>& slabs, std::vector<float>& acts,
                  std::vector<float>& dacts) {
      ga_update_args a = n.args(S, mb, perm.data(), kind);
      a.partials = part.data(); a.partials_floats = need;
      a.slabs = slabs.data(); a.max_splits = splits;
      a.acts = acts.data(); a.dacts = dacts.data();
      n.d.act_off[0] = 0; n.d.act_off[1] = mb * 256;
      return a;
    };
    ga_update_args a = mk(pol, pp, 0, partp, slp, ap, dp);
    ga_update_args b = mk(vf, pv, 1, partv, slv, av, dv);
    std::vector<double> ws_b(2048, 0.0);
    b.workspace = ws_b.data();
    a.step0 = 10; b.step0 = 20;
    g_log.clear();
    {
      const int rc8 = ga_update_epoch_pair(&a, (void*)0x10, &b, (void*)0x20);
      if (rc8) fprintf(stderr, "8b: rc %d error '%s'\n", rc8, g_error.c_str());
      if (getenv("GA_HARNESS_DUMP"))
        for (auto& l : g_log) fprintf(stderr, "  | %s\n", l.c_str());
      CHECK(rc8 == 0);
    }
    CHECK(count("pair_fwd stream=0x10") == 4 && count("pair_wgrad stream=0x10") == 4);
    CHECK(count("pair_dgrad stream=0x10") == 4 && count("pair_reduce stream=0x10") == 4);
    CHECK(count("pair_fwd stream=0x10 M=300 width=256 K=256 A=6/1 in=17/17") == 3);
    CHECK(count("pair_fwd stream=0x10 M=100") == 1);
    CHECK(count("pair_reduce stream=0x10 steps=11/21 regions=6/6 adam=1/1 M=300") == 1);
    CHECK(count("pair_reduce stream=0x10 steps=14/24") == 1);
    CHECK(count("fused_fwd") == 0 && count("reduce_regions") == 0);
    // stream_a first waits for what stream_b holds, stream_b then for the epoch
    CHECK(g_log.size() > 4 && g_log[0].rfind("record stream=0x20", 0) == 0 &&
          g_log[1].rfind("wait stream=0x10", 0) == 0 &&
          g_log[g_log.size() - 2].rfind("record stream=0x10", 0) == 0 &&
          g_log.back().rfind("wait stream=0x20", 0) == 0);
    // the two-stream schedule on request
    ga_set_merged_pair(0);
    g_log.clear();
    CHECK(ga_update_epoch_pair(&a, (void*)0x10, &b, (void*)0x20) == 0);
    CHECK(count("pair_") == 0 && count("fused_fwd") == 8 && count("reduce_regions n=") == 8);
    ga_set_merged_pair(1);
    // data parallel: never merged (each chain's all-reduce hides under the other)
    ga_set_allreduce_hook(fake_allreduce);
    a.comm = b.comm = (void*)0x1; a.world = b.world = 2; a.grad_scale = b.grad_scale = 0.5f;
    g_log.clear();
    CHECK(ga_update_epoch_pair(&a, (void*)0x10, &b, (void*)0x20) == 0);
    CHECK(count("pair_") == 0 && count("allreduce") == 8);
    a.comm = b.comm = nullptr;
    // unequal minibatch counts: not merged
    b.mb = 250;
    g_log.clear();
    CHECK(ga_update_epoch_pair(&a, (void*)0x10, &b, (void*)0x20) == 0);
    CHECK(count("pair_") == 0 && count("fused_fwd") == 8);
    ga_set_merged_pair(0);
  }
  // 9. Host-computed extents against what the kernels' index formulas reach, at the
  //    extreme shapes (32-wide nets, 1-row minibatches, 8 outputs, 32 inputs): every
  //    buffer is allocated at EXACTLY the size the Python side computes
  //    (ga_update_partials_floats; activation workspaces = largest minibatch x sum of
  //    the hidden widths; the flat parameter layout) and the fakes write / index the
  //    kernels' full extents, so an undersized buffer is an AddressSanitizer report.
  {
    ga_set_fused_train(1);
    ga_set_small_step(1);
    const int ins[] = {1, 4, 17, 31, 32};
    const int hs[] = {32, 64, 128, 256};
    const int outs[] = {1, 6, 8};
    const int64_t ms[] = {1, 31, 32, 63, 64, 65, 200, 4096};
    int fused = 0, narrow = 0, small = 0;
    for (int in : ins) for (int h : hs) for (int out : outs) for (int64_t M : ms) {
      Net net(in, h, h, out);
      const int64_t need = ga_update_partials_floats(&net.d, M);
      std::vector<float> partials((size_t)(need > 0 ? need : 1), 1.f);
      // activation workspaces as garage_amd/engine.py sizes them: rows x (h1 + h2)
      const int64_t ws_floats = M * 2 * (int64_t)h;
      std::vector<float> acts((size_t)ws_floats), dacts((size_t)ws_floats);
      std::vector<float> flat(net.params.size()), m1(net.params.size()),
          m2(net.params.size());
      std::vector<int32_t> perm((size_t)M);
      for (int64_t i = 0; i < M; ++i) perm[(size_t)i] = (int32_t)i;
      ga_update_args a = net.args(M, M, perm.data(), out == 1 ? 1 : 0);
      a.params = flat.data(); a.grads = flat.data();
      a.exp_avg = m1.data(); a.exp_avg_sq = m2.data();
      a.acts = acts.data(); a.dacts = dacts.data();
      net.d.act_off[0] = 0; net.d.act_off[1] = M * h;
      a.partials = need > 0 ? partials.data() : nullptr;
      a.partials_floats = need;
      const int64_t splits = ga_mlp_backward_splits(&net.d, M);
      std::vector<float> slabs((size_t)(splits * (int64_t)flat.size()));
      a.slabs = slabs.data(); a.max_splits = splits;
      g_ss_n_flat = (int64_t)flat.size();
      g_ss_ws_floats = ws_floats;
      g_log.clear();
      CHECK(ga_update_epoch(&a, nullptr) == 0);
      fused += count("fused_fwd");
      narrow += count("narrow M=");
      small += count("small_step");
      // a 2 x H net with <= 64 rows must only take the one-launch step when its two
      // exchanges fit the activation workspaces (32 rows up)
      if (count("small_step")) CHECK(M >= 32 && M <= 64);
    }
    g_ss_n_flat = 0;
    CHECK(g_ss_extent_errors == 0);
    CHECK(fused > 0 && narrow > 0 && small > 0);
  }
  if (g_failed) {
    fprintf(stderr, "%d check(s) failed\n", g_failed);
    return 1;
  }
  printf("host loops ok\n");
  return 0;
}
