// One optimisation epoch (all minibatches of one pass) enqueued from C++.
//
// VPG._train (torch/algos/vpg.py:230-248) iterates minibatches in Python and
// every minibatch costs ~14 kernel launches here.  These entry points walk the
// minibatches of one pass in native code (same kernels, same order: forward,
// fused loss + gradient seed, backward, slab reduction, [RCCL all-reduce],
// Adam), so the host cost is the HIP launch itself.
//
// ga_update_epoch_pair interleaves the policy pass and the value-function pass
// on two streams.  The reference runs them one after the other (vpg.py:244-248)
// but neither reads what the other writes, so the results are identical; on the
// GPU the two launch chains fill each other's latency bubbles (each GEMM of a
// 32768-row minibatch is a single wave of 512 workgroups).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/garage_amd.h"
#include "small_step.h"
#include "fused_train.h"

void ga_set_error(const char* fmt, ...);

typedef int (*ga_allreduce_fn)(void* comm, float* buf, int64_t n, void* stream);
static ga_allreduce_fn g_allreduce = nullptr;

extern "C" void ga_set_allreduce_hook(ga_allreduce_fn fn) { g_allreduce = fn; }

// Off by default: at the C3 minibatch the fused head + loss kernel takes what the
// narrow head GEMM and the loss kernel take together (18.8 us vs 11.7 + 8.3 us;
// one, two or four rows per 16-lane group and 16..64 rows per workgroup all land
// within 1 % end to end), so the older, simpler pair stays the default.
static int g_fuse_head = 0;
extern "C" int ga_set_fused_head_loss(int on) {
  g_fuse_head = on != 0;
  return 0;
}

// small_step.hip: one launch per optimizer step for minibatches of <= 64 rows
extern "C" int64_t ga_reduction_partials_doubles(void);

// gemm.hip: the backward pass of layers l_start .. 0 given d(loss)/d(pre-activation)
// of layer l_start in dacts; fused_first: the data gradient into layer 0's output
// and layer 0's weight gradient are computed elsewhere (ga_fused_dgrad_wgrad0)
extern "C" int ga_mlp_backward_range_f32(const ga_mlp_desc* d, const float* params,
                                         const float* X, int64_t ldx,
                                         const int32_t* row_idx, int64_t M,
                                         const float* acts, const float* dout,
                                         int64_t ldo, float* dacts, float* grad_slabs,
                                         int64_t slab_stride, int64_t n_splits,
                                         int l_start, int fused_first,
                                         ga_stream_t stream);

// The optimizer step with the streaming passes folded into the GEMM epilogues
// (fused_train.hip) for networks whose last hidden layer is 64 / 128 / 256 wide
// with a head of <= 8 outputs; 0 = the per-layer kernels (A/B runs, tests).
static int g_fused_train = 1;
extern "C" int ga_set_fused_train(int on) {
  g_fused_train = on != 0;
  return 0;
}

// GARAGE_AMD_TRACE=1: every launch of the epoch loop is announced on stderr and
// waited for (a faulting kernel is then the last one named)
static int trace_on() {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("GARAGE_AMD_TRACE");
    on = (e && e[0] == '1') ? 1 : 0;
  }
  return on;
}
#define GA_TRACE(stream, ...)                                   \
  do {                                                          \
    if (trace_on()) {                                           \
      (void)hipStreamSynchronize((hipStream_t)(stream));        \
      fprintf(stderr, "[ga_trace] " __VA_ARGS__);               \
      fprintf(stderr, "\n");                                    \
      fflush(stderr);                                           \
    }                                                           \
  } while (0)

// 2 x H networks with H = 32 or 64: forward + loss + backward in ONE launch
// (narrow_step.hip); 0 = the fused GEMM-epilogue kernels / per-layer kernels
static int g_narrow_step = 1;
extern "C" int ga_set_narrow_step(int on) {
  g_narrow_step = on != 0;
  return 0;
}

// The first layer's forward inside the fused last-hidden-layer kernel (3-layer
// networks with <= 32 inputs); 0 = its own streaming launch (A/B runs, tests)
// (GARAGE_AMD_FUSED_FIRST_LAYER=0 in the environment: the same, for A/B runs of
// programs that do not call the setter)
static int g_fused_first_layer = -1;
extern "C" int ga_set_fused_first_layer(int on) {
  g_fused_first_layer = on != 0;
  return 0;
}
static int fused_first_layer_on() {
  if (g_fused_first_layer < 0) {
    const char* e = getenv("GARAGE_AMD_FUSED_FIRST_LAYER");
    g_fused_first_layer = (e && e[0] == '0') ? 0 : 1;
  }
  return g_fused_first_layer;
}

namespace {
struct FusedPlan {
  bool ok = false;
  bool narrow = false;    // the whole step in one launch (narrow_step.hip)
  bool first = false;     // data gradient into layer 0 + its weight gradient fused too
  int64_t tiles = 0;
  int64_t lpart_off = 0, hpart_off = 0, wpart_off = 0, floats = 0;
  int64_t hstride = 0, wstride = 0;
};

// Layout of the partial-sum scratch for M rows (floats): [tiles][2] doubles of loss
// sums, [tiles][8 W + 8] head shares, [tiles][W1 ld0 + W1] first-layer shares.
FusedPlan fused_plan(const ga_mlp_desc* d, int64_t M) {
  FusedPlan f;
  const int L = d->n_layers;
  if (L < 2 || L > 8 || M < 1) return f;
  // the fused kernels implement tanh hidden layers and a linear output layer
  if (d->hidden_act != 0 || d->output_act != 0 || d->layer_norm) return f;
  if (g_narrow_step && ga_narrow_step_supported(L, d->dims)) {
    f.ok = f.narrow = true;
    f.tiles = ga_fused_tiles(M);
    f.hstride = ga_narrow_step_stride(d->dims[0], d->dims[1]);
    f.lpart_off = 0;
    f.hpart_off = 4 * f.tiles;
    f.wpart_off = f.hpart_off + f.tiles * f.hstride;
    f.floats = f.wpart_off;
    return f;
  }
  if (!ga_fused_width_ok(d->dims[L - 1]) || d->dims[L] > 8) return f;
  f.ok = true;
  f.tiles = ga_fused_tiles(M);
  f.first = L >= 3 && ga_fused_width_ok(d->dims[1]) && d->dims[0] <= 32;
  f.hstride = 8 * (int64_t)d->dims[L - 1] + 8;
  const int64_t ld0 = (d->dims[0] + 3) & ~3;
  f.wstride = f.first ? (int64_t)d->dims[1] * ld0 + d->dims[1] : 0;
  f.lpart_off = 0;
  f.hpart_off = 4 * f.tiles;
  f.wpart_off = f.hpart_off + f.tiles * f.hstride;
  f.floats = f.wpart_off + f.tiles * f.wstride;
  return f;
}
}  // namespace

extern "C" int64_t ga_update_partials_floats(const ga_mlp_desc* d, int64_t M) {
  if (!d) return 0;
  const FusedPlan f = fused_plan(d, M);
  return f.ok ? f.floats : 0;
}

static int g_small_step = 1;
extern "C" int ga_set_small_step(int on) {
  g_small_step = on != 0;
  return 0;
}

// ga_update_epoch_pair, data parallel: all-reduces of the two networks in one
// total order per GPU (see include/garage_amd.h)
static int g_ordered_allreduce = 1;
extern "C" int ga_set_ordered_allreduce(int on) {
  g_ordered_allreduce = on != 0;
  return 0;
}

namespace {

// events chaining the two networks' all-reduces; created once per process
// (re-recorded every step: a stream wait captures the record it was issued after)
struct ArOrder {
  hipEvent_t wait_for = nullptr;  // recorded after the other network's last all-reduce
  hipEvent_t record = nullptr;    // recorded after this one
};
hipEvent_t g_ar_events[2] = {nullptr, nullptr};

bool ar_events_ready() {
  for (int i = 0; i < 2; ++i) {
    if (!g_ar_events[i] &&
        hipEventCreateWithFlags(&g_ar_events[i], hipEventDisableTiming) != hipSuccess) {
      g_ar_events[i] = nullptr;
      return false;
    }
  }
  return true;
}

int check_args(const ga_update_args* a) {
  if (!a || !a->desc || !a->params || !a->X || !a->workspace) {
    ga_set_error("ga_update_epoch: null pointer");
    return -1;
  }
  if (a->S <= 0 || (a->perm && a->mb <= 0 && a->n_mb <= 0) || a->n_mb < 0 ||
      (a->perm && a->n_mb > a->S)) {
    ga_set_error("ga_update_epoch: bad sizes");
    return -1;
  }
  return 0;
}

int64_t n_minibatches(const ga_update_args* a) {
  if (a->perm && a->n_mb > 0) return a->n_mb;
  const int64_t mb = a->perm ? a->mb : a->S;
  return (a->S + mb - 1) / mb;
}

// rows the caller sized the activation workspaces for (its largest minibatch)
int64_t workspace_rows(const ga_update_args* a) {
  if (!a->perm) return a->S;
  if (a->n_mb > 0) return (a->S + a->n_mb - 1) / a->n_mb;
  return a->mb < a->S ? a->mb : a->S;
}

// ids [start, start + M) of the pass's permutation form minibatch k
void minibatch_range(const ga_update_args* a, int64_t k, int64_t* start, int64_t* M) {
  if (a->perm && a->n_mb > 0) {  // even split: a common count on every rank
    *start = k * a->S / a->n_mb;
    *M = (k + 1) * a->S / a->n_mb - *start;
    return;
  }
  const int64_t mb = a->perm ? a->mb : a->S;
  *start = k * mb;
  *M = (*start + mb <= a->S) ? mb : a->S - *start;
}

// the exchange + optimizer tail of a data-parallel step: grads hold this rank's
// scaled gradient
int allreduce_and_adam(const ga_update_args* a, int64_t k, ga_stream_t stream,
                       const ArOrder* order) {
  if (!g_allreduce) {
    ga_set_error("ga_update_epoch: no all-reduce hook installed");
    return -1;
  }
  if (order && order->wait_for &&
      hipStreamWaitEvent((hipStream_t)stream, order->wait_for, 0) != hipSuccess) {
    ga_set_error("ga_update_epoch: hipStreamWaitEvent failed");
    return -2;
  }
  int rc = g_allreduce(a->comm, a->grads, a->n_flat, stream);
  if (rc) {
    ga_set_error("ga_update_epoch: all-reduce failed (%d)", rc);
    return -2;
  }
  if (order && order->record &&
      hipEventRecord(order->record, (hipStream_t)stream) != hipSuccess) {
    ga_set_error("ga_update_epoch: hipEventRecord failed");
    return -2;
  }
  return ga_adam_step_f32(a->params, a->grads, a->exp_avg, a->exp_avg_sq, a->n_flat,
                          a->step0 + k + 1, a->lr, a->beta1, a->beta2, a->eps, stream);
}

float step_scale(const ga_update_args* a, int64_t k) {
  if (!a->comm && a->phase != 1) return 1.f;
  return (a->grad_scales_host && a->perm && a->n_mb > 0) ? a->grad_scales_host[k]
                                                          : a->grad_scale;
}

// One optimizer step on the fused kernels: hidden layers 0 .. L-3 per layer, then
// last hidden layer + head + loss + gradient seed in one launch, the middle
// layers' backward GEMMs, the data gradient into the first hidden layer with the
// first layer's weight gradient in one launch, and one reduction + Adam launch.
int run_minibatch_fused(const ga_update_args* a, const FusedPlan& f, int64_t k,
                        int64_t M, const int32_t* idx, float* loss_slot,
                        ga_stream_t stream_, const ArOrder* order) {
  hipStream_t stream = (hipStream_t)stream_;
  const ga_mlp_desc* d = a->desc;
  const int L = d->n_layers;
  const int out_w = d->dims[L];
  const int64_t splits = ga_mlp_backward_splits(d, M);
  if (splits > a->max_splits) {
    ga_set_error("ga_update_epoch: slab workspace too small");
    return -1;
  }
  auto r4 = [](int v) { return (int64_t)((v + 3) & ~3); };
  int rc;
  if (f.narrow) {
    ga_fused_loss_args la;
    memset(&la, 0, sizeof(la));
    la.kind = a->kind; la.actions = a->actions; la.lda = a->lda; la.old_ll = a->old_ll;
    la.adv = a->adv; la.returns = a->returns; la.idx = idx; la.log_std = a->params;
    la.has_min = a->has_min; la.has_max = a->has_max; la.min_log_std = a->min_log_std;
    la.max_log_std = a->max_log_std; la.A = out_w; la.algo = a->algo; la.clip = a->clip;
    la.ent_coeff = a->ent_coeff; la.ent_flags = a->ent_flags;
    la.double_softmax = a->double_softmax;
    double* lpart = reinterpret_cast<double*>(a->partials + f.lpart_off);
    float* part = a->partials + f.hpart_off;
    const int H = d->dims[1];
    rc = ga_narrow_train_step(a->params, d->w_off, d->b_off, d->dims[0], H, out_w, a->X,
                              a->ldx, M, &la, part, lpart, stream);
    if (rc) return rc;
    const int64_t ld0 = r4(d->dims[0]);
    const int64_t off[6] = {0, (int64_t)H * ld0, (int64_t)H * ld0 + H,
                            (int64_t)H * ld0 + H + (int64_t)H * H,
                            (int64_t)H * ld0 + 2 * H + (int64_t)H * H,
                            (int64_t)H * ld0 + 2 * H + (int64_t)H * H + 8 * (int64_t)H};
    ga_fused_region reg[6];
    for (int l = 0; l < 3; ++l) {
      reg[2 * l].beg = d->w_off[l];
      reg[2 * l].n = (int64_t)d->dims[l + 1] * r4(d->dims[l]);
      reg[2 * l + 1].beg = d->b_off[l];
      reg[2 * l + 1].n = d->dims[l + 1];
    }
    for (int k = 0; k < 6; ++k) {
      reg[k].src = part + off[k];
      reg[k].stride = f.hstride;
      reg[k].n_part = (int)f.tiles;
    }
    const bool exchange = a->comm && a->phase != 1;
    const bool do_adam = !exchange && a->phase != 1;
    rc = ga_reduce_regions_adam(reg, 6, a->params, a->grads, a->exp_avg, a->exp_avg_sq,
                                a->step0 + k + 1, a->lr, a->beta1, a->beta2, a->eps,
                                step_scale(a, k), do_adam ? 1 : 0, !a->learn_std, lpart,
                                (int)f.tiles, M, &la, loss_slot, stream);
    if (rc || !exchange) return rc;
    return allreduce_and_adam(a, k, stream_, order);
  }
  const bool first_in_kernel =
      fused_first_layer_on() && L == 3 && ga_fused_first_layer_ok(d->dims[0], d->dims[1]);
  if (L >= 3 && !first_in_kernel) {
    ga_mlp_desc below = *d;  // layers 0 .. L-3: the hidden layers under the last one
    below.n_layers = L - 1;
    rc = ga_mlp_forward_f32(&below, a->params, a->X, a->ldx, idx, M, a->acts, nullptr,
                            a->ldo, stream_);
    if (rc) return rc;
  }
  ga_fused_loss_args la;
  memset(&la, 0, sizeof(la));
  la.kind = a->kind; la.actions = a->actions; la.lda = a->lda; la.old_ll = a->old_ll;
  la.adv = a->adv; la.returns = a->returns; la.idx = idx; la.log_std = a->params;
  la.has_min = a->has_min; la.has_max = a->has_max; la.min_log_std = a->min_log_std;
  la.max_log_std = a->max_log_std; la.A = out_w; la.algo = a->algo; la.clip = a->clip;
  la.ent_coeff = a->ent_coeff; la.ent_flags = a->ent_flags;
  la.double_softmax = a->double_softmax;
  double* lpart = reinterpret_cast<double*>(a->partials + f.lpart_off);
  float* hpart = a->partials + f.hpart_off;
  float* wpart = a->partials + f.wpart_off;
  const int wl = d->dims[L - 1];  // last hidden width
  const float* Ain = L >= 3 ? a->acts + d->act_off[L - 3] : a->X;
  ga_fused_first_layer fl;
  if (first_in_kernel) {
    fl.X = a->X; fl.ldx = a->ldx; fl.W = a->params + d->w_off[0];
    fl.b = a->params + d->b_off[0]; fl.in_w = d->dims[0];
    fl.H = a->acts + d->act_off[0]; fl.ldh = r4(d->dims[1]);
  }
  rc = ga_fused_fwd_head_loss(Ain, L >= 3 ? r4(d->dims[L - 2]) : a->ldx,
                              L >= 3 ? nullptr : idx, a->params + d->w_off[L - 2],
                              r4(d->dims[L - 2]), a->params + d->b_off[L - 2], M, wl,
                              d->dims[L - 2], a->params + d->w_off[L - 1], r4(wl),
                              a->params + d->b_off[L - 1], &la,
                              a->dacts + d->act_off[L - 2], r4(wl), hpart, lpart,
                              first_in_kernel ? &fl : nullptr, stream);
  if (rc) return rc;
  rc = ga_mlp_backward_range_f32(d, a->params, a->X, a->ldx, idx, M, a->acts, nullptr,
                                 a->ldo, a->dacts, a->slabs, a->n_flat, splits, L - 2,
                                 f.first ? 1 : 0, stream);
  if (rc) return rc;
  if (f.first) {
    rc = ga_fused_dgrad_wgrad0(a->dacts + d->act_off[1], r4(d->dims[2]),
                               a->params + d->w_off[1], r4(d->dims[1]), M, d->dims[1],
                               d->dims[2], a->acts + d->act_off[0], r4(d->dims[1]), a->X,
                               a->ldx, idx, d->dims[0], wpart, stream);
    if (rc) return rc;
  }
  ga_fused_region reg[16];
  int nr = 0;
  for (int l = 0; l < L; ++l) {
    const int64_t wn = (int64_t)d->dims[l + 1] * r4(d->dims[l]);
    ga_fused_region& w = reg[nr++];
    ga_fused_region& b = reg[nr++];
    w.beg = d->w_off[l]; w.n = wn;
    b.beg = d->b_off[l]; b.n = d->dims[l + 1];
    if (l == L - 1) {
      w.src = hpart; b.src = hpart + 8 * (int64_t)wl;
      w.stride = b.stride = f.hstride; w.n_part = b.n_part = (int)f.tiles;
    } else if (l == 0 && f.first) {
      w.src = wpart; b.src = wpart + wn;
      w.stride = b.stride = f.wstride; w.n_part = b.n_part = (int)f.tiles;
    } else {
      w.src = a->slabs + d->w_off[l]; b.src = a->slabs + d->b_off[l];
      w.stride = b.stride = a->n_flat; w.n_part = b.n_part = (int)splits;
    }
  }
  const bool exchange = a->comm && a->phase != 1;
  const bool do_adam = !exchange && a->phase != 1;
  // (split-operand experiment: the optimizer launch rewrites the planes of the last
  // hidden layer's weights, the next step's forward launch then needs no plane launch)
  if (f.first && L == 3 && do_adam && ga_split_bf16_any() && d->dims[2] == 256 &&
      d->dims[1] % 32 == 0)
    ga_reduce_planes_hint(d->w_off[1], d->dims[2], d->dims[1]);
  rc = ga_reduce_regions_adam(reg, nr, a->params, a->grads, a->exp_avg, a->exp_avg_sq,
                              a->step0 + k + 1, a->lr, a->beta1, a->beta2, a->eps,
                              step_scale(a, k), do_adam ? 1 : 0, !a->learn_std, lpart,
                              (int)f.tiles, M, &la, loss_slot, stream);
  if (rc || !exchange) return rc;
  return allreduce_and_adam(a, k, stream_, order);
}

int run_minibatch(const ga_update_args* a, int64_t k, ga_stream_t stream,
                  const ArOrder* order = nullptr) {
  int64_t start, M;
  minibatch_range(a, k, &start, &M);
  const int L = a->desc->n_layers;
  const int out_w = a->desc->dims[L];
  const int32_t* idx = a->perm ? a->perm + start : nullptr;
  const int64_t splits = ga_mlp_backward_splits(a->desc, M);
  if (splits > a->max_splits) {
    ga_set_error("ga_update_epoch: slab workspace too small");
    return -1;
  }
  float* loss_slot = a->losses ? a->losses + k : a->loss_scratch;
  // a minibatch of <= 64 rows through a 2 x H net: the whole step in one launch
  // (the activation workspaces, unused on that path, carry its two exchanges:
  // they hold min(S, mb) x 2H floats each, enough from 32 rows up)
  if (g_small_step && !g_fuse_head && !a->comm && a->phase != 1 && a->kind >= 0 && a->kind <= 2 &&
      (a->algo == 0 || a->algo == 1) && a->acts && a->dacts &&
      workspace_rows(a) >= 32 && a->desc->hidden_act == 0 && a->desc->output_act == 0 &&
      !a->desc->layer_norm &&
      ga_small_step_supported(L, a->desc->dims, M) &&
      ga_small_step_resident(a->desc->dims[1], 2)) {
    ga_small_step_args s;
    memset(&s, 0, sizeof(s));
    s.params = a->params; s.exp_avg = a->exp_avg; s.exp_avg_sq = a->exp_avg_sq;
    for (int l = 0; l < 3; ++l) {
      s.w_off[l] = a->desc->w_off[l];
      s.b_off[l] = a->desc->b_off[l];
    }
    s.in_w = a->desc->dims[0]; s.H = a->desc->dims[1]; s.out_w = out_w; s.M = (int)M;
    s.X = a->X; s.ldx = a->ldx; s.idx = idx; s.kind = a->kind;
    s.double_softmax = a->double_softmax;
    s.actions = a->actions; s.lda = a->lda; s.old_ll = a->old_ll; s.adv = a->adv;
    s.returns = a->returns; s.algo = a->algo; s.clip = a->clip;
    s.has_min = a->has_min; s.has_max = a->has_max; s.min_log_std = a->min_log_std;
    s.max_log_std = a->max_log_std;
    s.ent_coeff = a->ent_coeff; s.ent_flags = a->ent_flags;
    s.step = a->step0 + k + 1; s.lr = a->lr; s.beta1 = a->beta1; s.beta2 = a->beta2;
    s.eps = a->eps; s.learn_std = a->learn_std;
    s.xh2 = a->acts; s.xdz = a->dacts;
    double* tail = a->workspace + ga_reduction_partials_doubles();
    s.bar = reinterpret_cast<unsigned*>(tail);
    s.fault = reinterpret_cast<int*>(tail + 1);
    s.loss_out = loss_slot;
    GA_TRACE(stream, "small_step k=%lld M=%lld H=%d in=%d out=%d kind=%d", (long long)k,
             (long long)M, s.H, s.in_w, s.out_w, s.kind);
    return ga_small_step(&s, stream);
  }
  if (g_fused_train && !g_fuse_head && a->partials && a->kind >= 0 && a->kind <= 2 &&
      (a->algo == 0 || a->algo == 1) && a->acts && a->dacts) {
    const FusedPlan f = fused_plan(a->desc, M);
    if (f.ok && f.floats <= a->partials_floats)
      return run_minibatch_fused(a, f, k, M, idx, loss_slot, stream, order);
  }
  // the head layer (hidden -> means / value) is computed inside the loss kernel
  // when its shape allows: no narrow GEMM launch, no round trip of its output
  const int hid_w = L >= 2 ? a->desc->dims[L - 1] : 0;
  const bool fuse_head = g_fuse_head && L >= 2 && (a->kind == 0 || a->kind == 1) &&
                         a->desc->output_act == 0 && !a->desc->layer_norm &&
                         ga_head_loss_supported(hid_w, out_w);
  GA_TRACE(stream, "forward k=%lld M=%lld start=%lld kind=%d", (long long)k, (long long)M,
           (long long)start, a->kind);
  int rc = ga_mlp_forward_f32(a->desc, a->params, a->X, a->ldx, idx, M, a->acts,
                              fuse_head ? nullptr : a->out, a->ldo, stream);
  if (rc) return rc;
  GA_TRACE(stream, "loss");
  const float* H = fuse_head ? a->acts + a->desc->act_off[L - 2] : nullptr;
  const int64_t ldh = (hid_w + 3) & ~3;
  const float* Wh = a->params + a->desc->w_off[L - 1];
  const float* bh = a->params + a->desc->b_off[L - 1];
  if (fuse_head && a->kind == 0) {
    rc = ga_head_ppo_gaussian_loss_f32(
        H, ldh, Wh, ldh, bh, hid_w, nullptr, 0, a->actions, a->lda, a->old_ll, a->adv,
        idx, a->params, a->has_min, a->min_log_std, a->has_max, a->max_log_std, M,
        out_w, a->algo, a->clip, a->ent_coeff, a->ent_flags, a->dout, a->ldo, nullptr,
        loss_slot, a->slabs, a->n_flat, splits, a->workspace, stream);
  } else if (fuse_head) {
    rc = ga_head_gaussian_nll_loss_f32(H, ldh, Wh, bh, hid_w, nullptr, 0, a->returns,
                                       idx, a->params, M, a->dout, a->ldo, loss_slot,
                                       a->slabs, a->n_flat, splits, a->workspace,
                                       stream);
  } else if (a->kind == 0) {
    rc = ga_ppo_gaussian_loss_f32(
        a->out, a->ldo, a->actions, a->lda, a->old_ll, a->adv, idx, a->params,
        a->has_min, a->min_log_std, a->has_max, a->max_log_std, M, out_w, a->algo,
        a->clip, a->ent_coeff, a->ent_flags, a->dout, nullptr, loss_slot, a->slabs,
        a->n_flat, splits, a->workspace, stream);
  } else if (a->kind == 2) {
    rc = ga_ppo_categorical_loss_f32(
        a->out, a->ldo, a->actions, a->lda, a->old_ll, a->adv, idx, M, out_w,
        a->double_softmax, a->algo, a->clip, a->ent_coeff, a->ent_flags, a->dout,
        nullptr, nullptr, loss_slot, nullptr, a->slabs, a->n_flat, splits,
        a->workspace, stream);
  } else {
    rc = ga_gaussian_nll_loss_f32(a->out, a->ldo, a->returns, idx, a->params, M,
                                  a->dout, loss_slot, a->slabs, a->n_flat, splits,
                                  a->workspace, stream);
  }
  if (rc) return rc;
  if (a->desc->output_act) {
    // output_nonlinearity: d(loss)/d(output) -> d(loss)/d(pre-activation)
    rc = ga_act_slope_mul_f32(a->dout, a->ldo, a->out, a->ldo, M, out_w,
                              a->desc->output_act, stream);
    if (rc) return rc;
  }
  GA_TRACE(stream, "backward splits=%lld", (long long)splits);
  rc = ga_mlp_backward_f32(a->desc, a->params, a->X, a->ldx, idx, M, a->acts,
                           a->dout, a->ldo, a->dacts, a->slabs, a->n_flat, splits,
                           stream);
  if (rc) return rc;
  GA_TRACE(stream, "reduce + adam");
  if (!a->comm && a->phase != 1) {
    // single process: slab sum and Adam in one launch
    return ga_reduce_adam_f32(a->slabs, splits, a->n_flat, a->params, a->grads,
                              a->exp_avg, a->exp_avg_sq, a->n_flat, a->step0 + k + 1,
                              a->lr, a->beta1, a->beta2, a->eps, !a->learn_std,
                              stream);
  }
  // data parallel: the global gradient is the sample-count weighted sum of the
  // rank gradients (mean over the union of the shards)
  rc = ga_reduce_slabs_f32(a->slabs, splits, a->n_flat, a->n_flat, step_scale(a, k),
                           a->grads, stream);
  if (rc) return rc;
  if (!a->learn_std) {
    if (hipMemsetAsync(a->grads, 0, sizeof(float), (hipStream_t)stream) !=
        hipSuccess) {
      ga_set_error("ga_update_epoch: memset failed");
      return -2;
    }
  }
  if (a->phase == 1) return 0;  // the caller exchanges and steps
  return allreduce_and_adam(a, k, stream, order);
}

}  // namespace

// Host-only: which ids of a pass's permutation form minibatch k, and how many
// minibatches the pass has -- the split ga_update_epoch* walks.  Exists so that
// the Python side's OptimizerWrapper.minibatch_bounds (which sizes workspaces and
// per-step gradient weights for data-parallel runs) can be checked against it
// without a GPU.  Returns the number of minibatches, < 0 on bad arguments.
extern "C" int64_t ga_minibatch_range(int64_t S, int64_t mb, int64_t n_mb, int has_perm,
                                      int64_t k, int64_t* start, int64_t* M) {
  ga_update_args a;
  memset(&a, 0, sizeof(a));
  a.S = S; a.mb = mb; a.n_mb = n_mb;
  a.perm = has_perm ? reinterpret_cast<const int32_t*>(&a) : nullptr;  // never read
  if (S <= 0 || (has_perm && mb <= 0 && n_mb <= 0) || n_mb < 0 || (has_perm && n_mb > S)) {
    ga_set_error("ga_minibatch_range: bad sizes");
    return -1;
  }
  const int64_t n = n_minibatches(&a);
  if (k < 0 || k >= n) {
    ga_set_error("ga_minibatch_range: k out of range");
    return -1;
  }
  if (start && M) minibatch_range(&a, k, start, M);
  return n;
}

extern "C" int ga_update_epoch(const ga_update_args* a, ga_stream_t stream) {
  ga_planes_epoch_begin();
  int rc = check_args(a);
  if (rc) return rc;
  const int64_t n_mb = n_minibatches(a);
  for (int64_t k = 0; k < n_mb; ++k) {
    rc = run_minibatch(a, k, stream);
    if (rc) return rc;
  }
  return 0;
}

// 1 (opt-in; GARAGE_AMD_MERGED_PAIR=1): when both passes take the fused 256-wide
// kernels with the same shapes, ga_update_epoch_pair runs step k of BOTH networks as
// four launches on ONE stream -- each launch a grid over the tiles of both
// (fused_train.hip / gemm.hip *_pair kernels) -- instead of two free-running
// four-launch chains on two streams.  Same arithmetic per network (bit-identical
// results) and the same schedule every time, but measured SLOWER than the two
// streams (round 3, C3: 114.2 ms per iteration against 110.0-111.8 in the two-stream
// mode's slow phase regime, 106 in its fast one, 116.2 on one stream): a pair launch
// takes exactly twice a single one (114.0 us against 2 x 57.2), because a CU's
// throughput on these kernels does not improve with more co-resident tiles of the
// SAME kernel -- what the two streams overlap is DIFFERENT
// kernels.  0 (default): the two-stream schedule.
static int g_merged_pair = -1;
extern "C" int ga_set_merged_pair(int on) {
  g_merged_pair = on != 0;
  return 0;
}
static int merged_pair_on() {
  if (g_merged_pair < 0) {
    const char* e = getenv("GARAGE_AMD_MERGED_PAIR");
    g_merged_pair = (e && e[0] == '1') ? 1 : 0;
  }
  return g_merged_pair;
}

namespace {

struct MergedNet {
  ga_fused_loss_args la;
  ga_fused_first_layer fl;
  double* lpart;
  float* hpart;
  float* wpart;
  ga_fused_region reg[8];
  int nr;
};

// can step k of the two passes run as pair launches?
bool merged_ok(const ga_update_args* a, const ga_update_args* b, int64_t k) {
  const ga_update_args* two[2] = {a, b};
  int64_t M0 = 0, start;
  for (int i = 0; i < 2; ++i) {
    const ga_update_args* x = two[i];
    int64_t M;
    minibatch_range(x, k, &start, &M);
    if (i == 0) M0 = M;
    if (M != M0 || M <= 64) return false;
    const ga_mlp_desc* d = x->desc;
    if (x->comm || x->phase != 0 || !g_fused_train || g_fuse_head || !x->partials ||
        x->kind < 0 || x->kind > 2 || (x->algo != 0 && x->algo != 1) || !x->acts ||
        !x->dacts || d->n_layers != 3)
      return false;
    const FusedPlan f = fused_plan(d, M);
    if (!f.ok || f.narrow || !f.first || f.floats > x->partials_floats) return false;
    if (!fused_first_layer_on() || !ga_fused_first_layer_ok(d->dims[0], d->dims[1]))
      return false;
    // (the pair kernels are compiled for two 256-wide hidden layers)
    if (!ga_fused_pair_supported(d->dims[2], d->dims[1], d->dims[0]) || d->dims[1] != 256)
      return false;
    if (d->dims[0] != a->desc->dims[0] || d->dims[1] != a->desc->dims[1] ||
        d->dims[2] != a->desc->dims[2])
      return false;
    if (ga_mlp_backward_splits(d, M) > x->max_splits) return false;
  }
  return ga_mlp_backward_splits(a->desc, M0) == ga_mlp_backward_splits(b->desc, M0);
}

void merged_fill(const ga_update_args* a, const FusedPlan& f, const int32_t* idx,
                 int64_t splits, MergedNet* n) {
  const ga_mlp_desc* d = a->desc;
  auto r4 = [](int v) { return (int64_t)((v + 3) & ~3); };
  const int out_w = d->dims[3];
  ga_fused_loss_args& la = n->la;
  memset(&la, 0, sizeof(la));
  la.kind = a->kind; la.actions = a->actions; la.lda = a->lda; la.old_ll = a->old_ll;
  la.adv = a->adv; la.returns = a->returns; la.idx = idx; la.log_std = a->params;
  la.has_min = a->has_min; la.has_max = a->has_max; la.min_log_std = a->min_log_std;
  la.max_log_std = a->max_log_std; la.A = out_w; la.algo = a->algo; la.clip = a->clip;
  la.ent_coeff = a->ent_coeff; la.ent_flags = a->ent_flags;
  la.double_softmax = a->double_softmax;
  n->lpart = reinterpret_cast<double*>(a->partials + f.lpart_off);
  n->hpart = a->partials + f.hpart_off;
  n->wpart = a->partials + f.wpart_off;
  ga_fused_first_layer& fl = n->fl;
  fl.X = a->X; fl.ldx = a->ldx; fl.W = a->params + d->w_off[0];
  fl.b = a->params + d->b_off[0]; fl.in_w = d->dims[0];
  fl.H = a->acts + d->act_off[0]; fl.ldh = r4(d->dims[1]);
  // the parameter regions, exactly as run_minibatch_fused lists them
  const int wl = d->dims[2];
  n->nr = 0;
  for (int l = 0; l < 3; ++l) {
    const int64_t wn = (int64_t)d->dims[l + 1] * r4(d->dims[l]);
    ga_fused_region& w = n->reg[n->nr++];
    ga_fused_region& bb = n->reg[n->nr++];
    w.beg = d->w_off[l]; w.n = wn;
    bb.beg = d->b_off[l]; bb.n = d->dims[l + 1];
    if (l == 2) {
      w.src = n->hpart; bb.src = n->hpart + 8 * (int64_t)wl;
      w.stride = bb.stride = f.hstride; w.n_part = bb.n_part = (int)f.tiles;
    } else if (l == 0) {
      w.src = n->wpart; bb.src = n->wpart + wn;
      w.stride = bb.stride = f.wstride; w.n_part = bb.n_part = (int)f.tiles;
    } else {
      w.src = a->slabs + d->w_off[l]; bb.src = a->slabs + d->b_off[l];
      w.stride = bb.stride = a->n_flat; w.n_part = bb.n_part = (int)splits;
    }
  }
}

// step k of both passes: four pair launches on `stream_`
int run_minibatch_merged(const ga_update_args* a, const ga_update_args* b, int64_t k,
                         ga_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  auto r4 = [](int v) { return (int64_t)((v + 3) & ~3); };
  int64_t sa, sb, M, Mb;
  minibatch_range(a, k, &sa, &M);
  minibatch_range(b, k, &sb, &Mb);
  const ga_mlp_desc* da = a->desc;
  const ga_mlp_desc* db = b->desc;
  const int64_t splits = ga_mlp_backward_splits(da, M);
  const FusedPlan fa = fused_plan(da, M), fb = fused_plan(db, M);
  const int32_t* ia = a->perm ? a->perm + sa : nullptr;
  const int32_t* ib = b->perm ? b->perm + sb : nullptr;
  MergedNet na, nb;
  merged_fill(a, fa, ia, splits, &na);
  merged_fill(b, fb, ib, splits, &nb);
  const int in_w = da->dims[0], K = da->dims[1], wl = da->dims[2];
  GA_TRACE(stream_, "merged step k=%lld M=%lld", (long long)k, (long long)M);
  int rc = ga_fused_fwd_head_loss_pair(
      M, wl, K,
      a->params + da->w_off[1], r4(K), a->params + da->b_off[1],
      a->params + da->w_off[2], r4(wl), a->params + da->b_off[2], &na.la,
      a->dacts + da->act_off[1], r4(wl), na.hpart, na.lpart, &na.fl,
      b->params + db->w_off[1], r4(K), b->params + db->b_off[1],
      b->params + db->w_off[2], r4(wl), b->params + db->b_off[2], &nb.la,
      b->dacts + db->act_off[1], r4(wl), nb.hpart, nb.lpart, &nb.fl, stream);
  if (rc) return rc;
  rc = ga_wgrad_mid_pair(M, splits, wl, K,
                         a->dacts + da->act_off[1], a->acts + da->act_off[0],
                         a->slabs + da->w_off[1], a->slabs + da->b_off[1], a->n_flat,
                         b->dacts + db->act_off[1], b->acts + db->act_off[0],
                         b->slabs + db->w_off[1], b->slabs + db->b_off[1], b->n_flat,
                         stream);
  if (rc) return rc;
  rc = ga_fused_dgrad_wgrad0_pair(
      M, K, wl, in_w,
      a->dacts + da->act_off[1], r4(wl), a->params + da->w_off[1], r4(K),
      a->acts + da->act_off[0], r4(K), a->X, a->ldx, ia, na.wpart,
      b->dacts + db->act_off[1], r4(wl), b->params + db->w_off[1], r4(K),
      b->acts + db->act_off[0], r4(K), b->X, b->ldx, ib, nb.wpart, stream);
  if (rc) return rc;
  ga_reduce_net ra, rb;
  const ga_update_args* two[2] = {a, b};
  MergedNet* nets[2] = {&na, &nb};
  const FusedPlan* plans[2] = {&fa, &fb};
  ga_reduce_net* rr[2] = {&ra, &rb};
  for (int i = 0; i < 2; ++i) {
    const ga_update_args* x = two[i];
    ga_reduce_net& r = *rr[i];
    r.regions = nets[i]->reg; r.n_regions = nets[i]->nr;
    r.params = x->params; r.grads = x->grads; r.exp_avg = x->exp_avg;
    r.exp_avg_sq = x->exp_avg_sq; r.step = x->step0 + k + 1; r.lr = x->lr;
    r.beta1 = x->beta1; r.beta2 = x->beta2; r.eps = x->eps; r.scale = step_scale(x, k);
    r.do_adam = 1; r.zero_slot0 = !x->learn_std; r.lpart = nets[i]->lpart;
    r.n_lpart = (int)plans[i]->tiles; r.M = M; r.loss = &nets[i]->la;
    r.loss_out = x->losses ? x->losses + k : x->loss_scratch;
  }
  return ga_reduce_regions_adam_pair(&ra, &rb, stream);
}

hipEvent_t g_merge_events[2] = {nullptr, nullptr};

}  // namespace

extern "C" int ga_update_epoch_pair(const ga_update_args* a, ga_stream_t stream_a,
                                    const ga_update_args* b, ga_stream_t stream_b) {
  ga_planes_epoch_begin();
  int rc = check_args(a);
  if (rc) return rc;
  rc = check_args(b);
  if (rc) return rc;
  if (a->params == b->params || a->slabs == b->slabs || a->workspace == b->workspace ||
      a->acts == b->acts) {
    ga_set_error("ga_update_epoch_pair: the two passes must not share buffers");
    return -1;
  }
  const int64_t na = n_minibatches(a), nb = n_minibatches(b);
  // (the pair kernels have no split-operand instantiation: with that experiment on the
  // two-stream schedule runs, so that one switch decides the arithmetic of every step)
  if (merged_pair_on() && na == nb && !ga_split_bf16_any()) {
    bool all = true;
    for (int64_t k = 0; k < na && all; ++k) all = merged_ok(a, b, k);
    if (all) {
      // everything goes to stream_a; stream_b is ordered around it, so that work the
      // caller enqueues on either stream before / after this epoch sees the same
      // dependencies as with the two-stream schedule
      for (int i = 0; i < 2; ++i)
        if (!g_merge_events[i] &&
            hipEventCreateWithFlags(&g_merge_events[i], hipEventDisableTiming) !=
                hipSuccess) {
          g_merge_events[i] = nullptr;
          ga_set_error("ga_update_epoch_pair: cannot create events");
          return -2;
        }
      if (stream_a != stream_b) {
        if (hipEventRecord(g_merge_events[0], (hipStream_t)stream_b) != hipSuccess ||
            hipStreamWaitEvent((hipStream_t)stream_a, g_merge_events[0], 0) != hipSuccess) {
          ga_set_error("ga_update_epoch_pair: stream ordering failed");
          return -2;
        }
      }
      for (int64_t k = 0; k < na; ++k) {
        rc = run_minibatch_merged(a, b, k, stream_a);
        if (rc) return rc;
      }
      if (stream_a != stream_b) {
        if (hipEventRecord(g_merge_events[1], (hipStream_t)stream_a) != hipSuccess ||
            hipStreamWaitEvent((hipStream_t)stream_b, g_merge_events[1], 0) != hipSuccess) {
          ga_set_error("ga_update_epoch_pair: stream ordering failed");
          return -2;
        }
      }
      return 0;
    }
  }
  ArOrder oa, ob;
  if (g_ordered_allreduce && a->comm && b->comm) {
    if (!ar_events_ready()) {
      ga_set_error("ga_update_epoch_pair: cannot create events");
      return -2;
    }
    oa.wait_for = g_ar_events[1]; oa.record = g_ar_events[0];
    ob.wait_for = g_ar_events[0]; ob.record = g_ar_events[1];
  }
  for (int64_t k = 0; k < (na > nb ? na : nb); ++k) {
    if (k < na) {
      rc = run_minibatch(a, k, stream_a, &oa);
      if (rc) return rc;
    }
    if (k < nb) {
      rc = run_minibatch(b, k, stream_b, &ob);
      if (rc) return rc;
    }
  }
  return 0;
}
