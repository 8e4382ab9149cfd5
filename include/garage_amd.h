/* garage_amd -- C ABI of the MI355X (gfx950) on-policy rollout + PPO update engine.
 *
 * The reference (akolobov/garage v2021.03.0) is 100 % Python and has no FFI: its
 * plugin surface is a set of Python classes.  This header
 * is the boundary between that Python surface (mirrored by the `garage_amd`
 * package) and the hand-written HIP kernels; each entry point names the
 * reference code it replaces (paths relative to /root/reference/src/garage).
 * INTEGRATION.md shows the ctypes binding a garage maintainer would add.
 *
 * Conventions
 *   - extern "C"; plain pointers and sizes; no torch types.
 *   - every pointer is a caller-owned DEVICE pointer unless the name ends in
 *     `_host`; fp32 unless noted; row-major; "ld*" = floats between rows.
 *   - matrices handed to the MLP entry points must be 16-byte aligned with
 *     leading dimensions that are multiples of 4 floats (garage_amd pads).
 *   - asynchronous on `stream` (a hipStream_t, passed as void*); no allocation,
 *     no synchronisation, graph-capture safe; scratch is passed in.
 *   - return 0 on success, <0 on error; ga_last_error() returns a thread-local
 *     message.  Nothing throws across the boundary.
 */
#ifndef GARAGE_AMD_H_
#define GARAGE_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ga_stream_t; /* hipStream_t */

int ga_abi_version(void); /* 4 */
/* Wherever a policy's scalar std parameter crosses this interface it comes as
 * (log_std pointer, has_min, min_log_std, has_max, max_log_std) and goes through
 * GaussianMLPBaseModule.forward's transformation
 * (torch/modules/gaussian_mlp_module.py:165-181): clamp to [min, max], then
 * std = exp(p) or, when bit 1 of `has_min` is set (std_parameterization =
 * 'softplus'), std = log(1 + exp(exp(p))).  Bit 0 of `has_min` = the lower clamp is
 * present.  The gradient written for the parameter includes the transformation's
 * slope (0 through an active clamp). */
const char* ga_last_error(void);

/* ---- returns + GAE(lambda) -------------------------------------------------
 * Replaces garage.np.discount_cumsum called per padded row
 * (np/_functions.py:111-128; call site torch/algos/vpg.py:149-153) and
 * garage.torch.compute_advantages (torch/_functions.py:25-85) including the
 * zero-padding semantics of short episodes.
 *   mode 0: rows are env slices of the (n_rows, T) rollout buffer, row stride
 *           ld; tail[i] = episode length at an episode's last step, else 0.
 *   mode 1: every row is one episode (packed batch via offsets[n_rows+1] with
 *           max_len >= the longest row -- a contract: rows are scanned in one
 *           pass of max_len steps when max_len <= 256 -- or a padded (N, P)
 *           batch with offsets = NULL).
 * v0 = value of the all-zero observation (content of padded baselines);
 * bonus / bonus_const = per-step / constant reward bonus added for the
 * advantages only (entropy_method='max', vpg.py:158-160).
 * Algorithmic HBM bytes: 16 per step. */
int ga_gae_scan_f32(const float* rewards, const float* values, const float* bonus,
                    const uint16_t* tail, const int64_t* offsets, int64_t n_rows,
                    int64_t T, int64_t ld, int64_t max_len, int mode,
                    int max_episode_length, double discount, double gae_lambda,
                    float v0, float bonus_const, float* adv, float* ret,
                    ga_stream_t stream);
/* Whole-episode rows (mode 1) of at most 256 steps without a per-step bonus take
 * constant-decay fast paths (fixed horizon: aligned rows of exactly
 * max_episode_length steps; otherwise the ragged variant); 0 forces the general
 * kernel (A/B runs, tests). */
int ga_set_gae_fixed_fast_path(int on);
/* Steps per lane of those fast paths: 4 (one 16-B access per array and lane) or 8
 * (two; half the lanes, waves and shuffle steps per row).  Same recurrences in
 * another association order: results agree to fp64 rounding. */
int ga_set_gae_rows_steps_per_lane(int steps);

/* ---- MLP forward / backward (fp32 MFMA GEMMs) -------------------------------
 * Replaces MLPModule / MultiHeadedMLPModule.forward
 * (torch/modules/multi_headed_mlp_module.py:136-151) and its autograd backward
 * as used by VPG._train_policy / _train_value_function (vpg.py:250-293).
 * Hidden activations: tanh (the GaussianMLP* defaults,
 * torch/policies/gaussian_mlp_policy.py:44-60), relu or none. */
typedef struct {
  int32_t n_layers;   /* linear layers incl. the output layer, 1..8 */
  int32_t dims[9];    /* dims[0] = input width, dims[l+1] = width of layer l */
  int64_t w_off[8];   /* W_l [dims[l+1]][round4(dims[l])] offset (floats) in params */
  int64_t b_off[8];   /* b_l [dims[l+1]] offset (floats) in params */
  int64_t act_off[8]; /* hidden layer l output offset in the activation workspace,
                         row stride round4(dims[l+1]) */
  int32_t hidden_act; /* hidden_nonlinearity of MLPModule
                         (torch/modules/mlp_module.py:43-44): 0 = tanh (the
                         GaussianMLP* default; what a zeroed descriptor means),
                         1 = relu, 2 = none.  The fused / one-launch kernels
                         implement tanh; other networks take the per-layer GEMMs. */
  int32_t output_act; /* output_nonlinearity of the same module applied to the last
                         layer (the Gaussian mean / the value): 0 = none (default),
                         1 = tanh, 2 = relu.  Per-layer kernels only; the caller
                         scales the loss's d(output) by the slope before the backward
                         pass (ga_act_slope_mul_f32). */
  int32_t layer_norm; /* layer_normalization of the same module
                         (multi_headed_mlp_module.py:77-81): LayerNorm(eps 1e-5,
                         affine) over the input of every hidden linear layer.
                         Per-layer kernels only. */
  int32_t pad_;
  int64_t ln_off[8];  /* gamma_l [round4(dims[l])] offset in params; beta_l follows */
  int64_t lnx_off[8]; /* normalised input of hidden layer l in the activation
                         workspace (row stride round4(dims[l])) ... */
  int64_t lns_off[8]; /* ... and its per-row (mean, rstd) pairs */
} ga_mlp_desc;

/* dout[i, j] *= slope of the activation `act` (1 tanh, 2 relu) at its OUTPUT
 * out[i, j], j < N: turns the loss's gradient with respect to an output layer
 * with an output_nonlinearity into the gradient with respect to its pre-activation
 * (what ga_mlp_backward_f32 takes). */
int ga_act_slope_mul_f32(float* dout, int64_t ldd, const float* out, int64_t ldo,
                         int64_t M, int N, int act, ga_stream_t stream);

/* out[M, ldo] = MLP(X[row_idx[i]] or X[i]); acts keeps the hidden outputs. */
int ga_mlp_forward_f32(const ga_mlp_desc* d, const float* params, const float* X,
                       int64_t ldx, const int32_t* row_idx, int64_t M, float* acts,
                       float* out, int64_t ldo, ga_stream_t stream);
/* The same forward with every layer in ONE launch (activations of 32 rows stay in
 * LDS between layers, weights stream from L2); ga_mlp_forward_f32 dispatches to
 * it when ga_policy_step_fused_supported(d).  ga_set_fused_forward(0) forces
 * the per-layer GEMM path (A/B measurements). */
int ga_mlp_forward_fused_f32(const ga_mlp_desc* d, const float* params,
                             const float* X, int64_t ldx, const int32_t* row_idx,
                             int64_t M, float* acts, float* out, int64_t ldo,
                             ga_stream_t stream);
int ga_set_fused_forward(int on);
/* Outputs only: ga_mlp_forward_f32 with acts == NULL computes the whole network in
 * one launch without writing any activation (two hidden tanh layers of 128 / 256
 * units, <= 32 inputs, <= 8 linear outputs: the full-batch evaluation passes of
 * vpg.py:147-184 -- baselines, old log-likelihoods, LossBefore / LossAfter / KL).
 * ga_mlp_forward_eval_supported(d) says whether a network qualifies;
 * ga_set_eval_forward(0) (GARAGE_AMD_EVAL_FORWARD=0) makes it answer no. */
int ga_mlp_forward_eval_supported(const ga_mlp_desc* d);
int ga_set_eval_forward(int on);
/* Layer products with one dimension <= 32 (first-layer forward, head data
 * gradient, first-layer / head weight gradients) run as HBM-streaming kernels
 * instead of MFMA tiles (default on; 0 = MFMA tiles everywhere, for A/B runs). */
int ga_set_skinny_kernels(int on);
/* The head layer's weight-gradient streaming kernel also writes the data gradient
 * of the layer below (same pass over the hidden activations) when the head is
 * <= 16 wide (default on; 0 = separate data-gradient launch, for A/B runs). */
int ga_set_fused_head_dgrad(int on);
/* ga_mlp_forward_f32 applies the head layer (<= 8 outputs) in the epilogue of the
 * last hidden layer's GEMM when that layer is 64 or 128 units wide (mode 2: also
 * 256) and M is a multiple of 64: a workgroup's tile then spans whole rows, so the
 * staged tanh outputs are multiplied with the head weights before they leave the
 * CU and the narrow head GEMM (one more pass over the hidden activations)
 * disappears.  mode 0 = separate head launch, 1 = default, 2 = 256-wide layers too
 * (faster with one update chain on the chip, slower with two; gemm.hip). */
int ga_set_fused_head_forward(int mode);
/* GEMMs with at most 128 tiles of 128 x 128 (minibatches of a few thousand rows
 * and below, e.g. the reference's default minibatch of 64) run on 64 x 64 tiles
 * with 128-deep k-steps: more workgroups, K / 128 dependent memory round trips
 * instead of K / 32; bit-identical results (default on; 0 for A/B runs). */
int ga_set_small_m_gemm(int on);
/* ga_update_epoch*: a minibatch of <= 64 rows through a network of two equal
 * tanh hidden layers (32, 64, ... 256 units, <= 32 inputs, <= 8 outputs;
 * Gaussian or categorical PPO / VPG objective with the entropy options, or the
 * value NLL; one process) takes its whole optimizer step -- gather, forward, loss, backward,
 * Adam -- in ONE launch of H / 16 workgroups with two grid barriers
 * (small_step.hip) instead of ten dependent launches.  Same formulas, other
 * summation orders: results agree to rounding.  The last slot of the reduction
 * workspace is raised when a barrier gave up (never seen); that launch and every
 * later one then leave the parameters untouched.  Default on; 0 = per-layer
 * path. */
int ga_set_small_step(int on);
/* Test hooks of that path.  resident_cap: workgroups the device is taken to hold
 * at once (< 0: ask the occupancy query; 0 forces the per-layer path -- the
 * fallback for shapes whose two concurrent grids would not be co-resident).
 * max_polls: polls after which a waiter at a grid barrier abandons the launch
 * (< 0: default 2^22; 0 forces the abort path).  An abandoned launch writes no
 * parameter; the fault word stays raised and later launches return at once until
 * the caller has cleared the last two workspace slots. */
int ga_set_small_step_resident_cap(int workgroups);
int ga_set_small_step_max_polls(int polls);
int64_t ga_small_step_launches(void); /* launches so far (tests, diagnostics) */
/* Developer hook (tools/small_step_phases.py): the first call arms the recording
 * and returns 1; later calls synchronise and copy 16 timestamps (100 MHz wall
 * clock of workgroup 0 at the phase boundaries of the most recent launch) to the
 * HOST array and return 0. */
int ga_small_step_debug(long long* host_out16);
/* the same for the one-launch step of 2 x 32 / 2 x 64 networks (narrow_step.hip) */
int ga_narrow_step_debug(long long* host_out16);
/* ... for the fused rollout step (policy_fused.hip) */
int ga_policy_step_debug(long long* host_out32);
/* ... and for the fused last-hidden-layer + head + loss kernel (fused_train.hip) */
int ga_fused_fwd_debug(long long* host_out16);
/* (start, end of the k-loop, end) of the first n <= 4096 workgroups of that launch */
int ga_fused_fwd_debug_skew(long long* host_out, int n);
/* the data-gradient + first-layer weight-gradient kernel: which = 0 -> 16 phase
 * stamps of one workgroup (the first call arms the hook and returns 1), which = 1
 * -> (start, end of the k-loop, end) of the first n workgroups */
int ga_fused_dgrad_debug(long long* host_out, int which, int n);
/* Forward-mode tangent of the MLP (torch/optimizers/conjugate_gradient_optimizer.py
 * :18-66 takes the same product by double backward): with dtheta = tangent (flat
 * parameter layout) and acts = the hidden activations of a forward at the same
 * rows, tout[M, ldo] = d(output); tacts is a workspace shaped like acts. */
int ga_mlp_jvp_f32(const ga_mlp_desc* d, const float* params, const float* tangent,
                   const float* X, int64_t ldx, const int32_t* row_idx, int64_t M,
                   const float* acts, float* tacts, float* tout, int64_t ldo,
                   ga_stream_t stream);
/* split count ga_mlp_backward_f32 should be called with for M rows */
int64_t ga_mlp_backward_splits(const ga_mlp_desc* d, int64_t M);
/* dout = dLoss/d(out).  Writes n_splits partial gradient slabs, each laid out
 * like `params` (slab_stride floats apart); ga_reduce_slabs_f32 sums them. */
int ga_mlp_backward_f32(const ga_mlp_desc* d, const float* params, const float* X,
                        int64_t ldx, const int32_t* row_idx, int64_t M,
                        const float* acts, const float* dout, int64_t ldo,
                        float* dacts, float* grad_slabs, int64_t slab_stride,
                        int64_t n_splits, ga_stream_t stream);
/* C[M,N] = A[M,K] B[N,K]^T, exposed for tests of the GEMM core. */
int ga_gemm_nt_f32(const float* A, int64_t lda, const float* B, int64_t ldb,
                   float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                   ga_stream_t stream);

/* ---- losses -----------------------------------------------------------------
 * workspace: ga_reduction_workspace_doubles() doubles of device scratch, ZEROED
 * once by the caller (its last two slots are a ticket / barrier pair every launch
 * leaves at 0 and a fault flag), used by one stream at a time. */
int64_t ga_reduction_workspace_doubles(void);
/* Loss launches of one block (<= 256 rows) write the loss and the log-std gradient
 * slot themselves.  1 = launches of several blocks do too: the block that draws
 * the last ticket adds the blocks' partial sums in block order (the bits of the
 * one-wave finalize launch it replaces); default 0 -- the device-scope fences cost
 * more than the launch they save (losses.hip). */
int ga_set_one_launch_losses(int on);

/* PPO clipped surrogate (algo 0; torch/algos/ppo.py:96-132) or VPG objective
 * (algo 1; vpg.py:434-454) for a Gaussian policy with a scalar log-std
 * (torch/modules/gaussian_mlp_module.py:158-192), + entropy regularisation
 * (vpg.py:343-345,408-432; ent_flags bit0 regularized, bit1 softplus, bit2
 * stop-gradient).  Writes loss = -mean(objective), optionally dLoss/dmean,
 * the new log-likelihoods and dLoss/dlog_std (into slot 0 of slab 0). */
int ga_ppo_gaussian_loss_f32(const float* mean, int64_t ldm, const float* actions,
                             int64_t lda, const float* old_ll, const float* adv,
                             const int32_t* idx, const float* log_std, int has_min,
                             float min_log_std, int has_max, float max_log_std,
                             int64_t M, int A, int algo, float clip, float ent_coeff,
                             int ent_flags, float* dmean, float* ll_out,
                             float* loss_out, float* grad_slab0, int64_t slab_stride,
                             int64_t n_splits, double* workspace, ga_stream_t stream);
/* The same objective for a categorical MLP head (discrete actions, BASELINE.json
 * configs 1-2).  No torch CategoricalMLPPolicy exists in the reference; the
 * convention follows its torch categorical policies, which pass
 * softmax(net(x)) as `logits=` (torch/policies/categorical_cnn_policy.py:138-139;
 * double_softmax = 1), and log_prob casts float actions with .long()
 * (SURVEY.md Q24).  ent_out / ent_sum_out: per-row / summed entropies. */
int ga_ppo_categorical_loss_f32(const float* scores, int64_t lds,
                                const float* actions, int64_t lda,
                                const float* old_ll, const float* adv,
                                const int32_t* idx, int64_t M, int A,
                                int double_softmax, int algo, float clip,
                                float ent_coeff, int ent_flags, float* dscores,
                                float* ll_out, float* ent_out, float* loss_out,
                                double* ent_sum_out, float* grad_slab0,
                                int64_t slab_stride, int64_t n_splits,
                                double* workspace, ga_stream_t stream);
int ga_categorical_kl_f32(const float* scores_old, const float* scores_new,
                          int64_t ld, int64_t M, int A, int double_softmax,
                          double* kl_sum_out, double* workspace, ga_stream_t stream);
/* GaussianMLPValueFunction.compute_loss
 * (torch/value_functions/gaussian_mlp_value_function.py:81-98). */
int ga_gaussian_nll_loss_f32(const float* v, int64_t ldv, const float* returns,
                             const int32_t* idx, const float* log_std, int64_t M,
                             float* dv, float* loss_out, float* grad_slab0,
                             int64_t slab_stride, int64_t n_splits, double* workspace,
                             ga_stream_t stream);
/* The head layer fused into the loss: H[M, ldh] are the last hidden activations,
 * W[A][ldw] / bias[A] the head's weights (torch/modules/gaussian_mlp_module.py
 * :288-305 output layer).  Same loss, gradient seed (dmean / dv, row stride ldd)
 * and log-std gradient as the unfused entry points above; mean_out / v_out
 * optionally receive the head outputs.  Supported shapes: ga_head_loss_supported.
 * ga_set_fused_head_loss(1) makes ga_update_epoch* use them (default 0: measured
 * equal to the head GEMM + loss kernel pair at the C3 minibatch). */
int ga_head_loss_supported(int hidden_width, int A);
int ga_set_fused_head_loss(int on);
int ga_head_ppo_gaussian_loss_f32(
    const float* H, int64_t ldh, const float* W, int64_t ldw, const float* bias,
    int hidden_width, float* mean_out, int64_t ldm, const float* actions, int64_t lda,
    const float* old_ll, const float* adv, const int32_t* idx, const float* log_std,
    int has_min, float min_log_std, int has_max, float max_log_std, int64_t M, int A,
    int algo, float clip, float ent_coeff, int ent_flags, float* dmean, int64_t ldd,
    float* ll_out, float* loss_out, float* grad_slab0, int64_t slab_stride,
    int64_t n_splits, double* workspace, ga_stream_t stream);
int ga_head_gaussian_nll_loss_f32(
    const float* H, int64_t ldh, const float* W, const float* bias, int hidden_width,
    float* v_out, int64_t ldv_out, const float* returns, const int32_t* idx,
    const float* log_std, int64_t M, float* dv, int64_t ldd, float* loss_out,
    float* grad_slab0, int64_t slab_stride, int64_t n_splits, double* workspace,
    ga_stream_t stream);
/* sum over rows of KL(old || new), VPG._compute_kl_constraint (vpg.py:381-406) */
int ga_gaussian_kl_f32(const float* mean_old, const float* mean_new, int64_t ld,
                       int64_t M, int A, float log_std_old, float log_std_new,
                       double* kl_sum_out, double* workspace, ga_stream_t stream);

/* ---- optimiser: OptimizerWrapper.step == torch.optim.Adam.step
 * (torch/optimizers/optimizer_wrapper.py:53-63, _functions.py:25-65) */
int ga_reduce_slabs_f32(const float* slabs, int64_t n_splits, int64_t slab_stride,
                        int64_t n, float scale, float* out, ga_stream_t stream);
/* ga_reduce_slabs_f32 (scale 1) + ga_adam_step_f32 in one launch; the reduced
 * gradient is also written to `grads`.  zero_slot0: keep element 0 (the log-std
 * slot) untrained. */
int ga_reduce_adam_f32(const float* slabs, int64_t n_splits, int64_t slab_stride,
                       float* params, float* grads, float* exp_avg,
                       float* exp_avg_sq, int64_t n, int64_t step, double lr,
                       double beta1, double beta2, double eps, int zero_slot0,
                       ga_stream_t stream);
int ga_adam_step_f32(float* params, const float* grads, float* exp_avg,
                     float* exp_avg_sq, int64_t n, int64_t step, double lr,
                     double beta1, double beta2, double eps, ga_stream_t stream);
/* make_optimizer (_functions.py:25-65) builds ANY torch.optim class; beyond the default
 * Adam (fused into the update kernels) these run as one elementwise launch over the
 * flat buffer, torch's single-tensor arithmetic:
 *   kind 1 torch.optim.SGD      h = {lr, momentum, dampening, weight_decay, -};
 *                               flags bit 0 nesterov; s1 = momentum buffer
 *   kind 2 torch.optim.RMSprop  h = {lr, alpha, eps, weight_decay, momentum}; flags bit 0
 *                               centered; s1 square_avg, s2 momentum buffer, s3 grad_avg
 *   kind 3 torch.optim.Adam / AdamW with weight_decay / amsgrad
 *                               h = {lr, beta1, beta2, eps, weight_decay}; flags bit 0
 *                               amsgrad, bit 1 decoupled decay (AdamW); s1 exp_avg,
 *                               s2 exp_avg_sq, s3 max_exp_avg_sq
 * h: HOST pointer to 5 doubles; step counts from 1; unused state pointers may be NULL. */
int ga_optimizer_step_f32(int kind, float* params, const float* grads, float* s1,
                          float* s2, float* s3, int64_t n, int64_t step, const double* h,
                          int flags, ga_stream_t stream);

/* ---- advantage centring: VPG._compute_advantage (vpg.py:371-377)
 * stats = device double[4]: sum, count, sum of squared deviations, min.
 * what: 0 -> stats[0..1], 1 -> stats[2] (uses the mean in stats), 2 -> stats[3]. */
/* ---- constrained (TRPO) policy step: vectors of n_flat floats -----------------
 * conjugate_gradient_optimizer.py:69-104,146-186.  ga_dot_f32: *out (device
 * double) = sum a[i] b[i], fp64 accumulation, fixed order.  ga_axpby_f32:
 * y = alpha x + beta y.  ga_fisher_seed_gaussian_f32: dout = tmean exp(-2 s) / M,
 * the Gaussian-mean block of the KL Hessian at old == new (s = clamped log-std). */
int ga_dot_f32(const float* a, const float* b, int64_t n, double* out,
               ga_stream_t stream);
int ga_axpby_f32(double alpha, const float* x, double beta, float* y, int64_t n,
                 ga_stream_t stream);
int ga_fisher_seed_gaussian_f32(const float* tmean, int64_t ldt, int64_t M, int A,
                                const float* log_std, int has_min, float min_log_std,
                                int has_max, float max_log_std, float* dout,
                                int64_t ldd, ga_stream_t stream);
/* The same seed for the categorical head (TRPO with CategoricalMLPPolicy): the KL
 * Hessian with respect to the class scores at old == new, applied to the scores'
 * tangent -- J1 (diag(q) - q q^T) J1 tscores / M with q the class probabilities and J1
 * the Jacobian of the inner softmax of the reference's head
 * (torch/policies/categorical_cnn_policy.py:138-139; identity when double_softmax = 0).
 * A <= 32. */
int ga_fisher_seed_categorical_f32(const float* scores, int64_t lds, const float* tscores,
                                   int64_t ldt, int64_t M, int A, int double_softmax,
                                   float* dout, int64_t ldd, ga_stream_t stream);
int ga_stats_f32(const float* x, int64_t n, int what, double* stats,
                 double* workspace, ga_stream_t stream);
int ga_adv_center_f32(float* x, int64_t n, const double* stats, float eps,
                      ga_stream_t stream);
int ga_sub_scalar_f32(float* x, int64_t n, const double* scalar, ga_stream_t stream);

/* ---- rollout ----------------------------------------------------------------
 * Synthetic batched environment (the benchmark workload of BASELINE.json;
 * Environment.reset/step contract of _environment.py:237-276). */
typedef struct {
  int64_t n;
  int64_t env_id0;
  int32_t obs_dim, act_dim, discrete, min_len, max_len;
  uint64_t seed;
  int32_t* episode; /* [n] */
  int32_t* t;       /* [n] */
  int32_t* len;     /* [n] */
} ga_synth_env;

int ga_synth_env_reset(const ga_synth_env* env, const uint8_t* mask, float* obs,
                       int64_t ldo, ga_stream_t stream);
int ga_synth_env_step(const ga_synth_env* env, const float* actions, int64_t lda,
                      const float* obs, float* next_obs, int64_t ldo, float* reward,
                      uint8_t* step_type, ga_stream_t stream);

/* NormalizedEnv's observation / reward normalisation
 * (envs/normalized_env.py:118-132,134-164): per-env float64 moving mean and
 * variance, updated before use; rows with mask == 0 (or all when NULL) only.
 * reward = (normalize ? r / (sqrt(var) + 1e-8) : r) * scale. */
int ga_obs_normalize_f64(int64_t n, int obs_dim, float* obs, int64_t ldo,
                         double* mean, double* var, double alpha,
                         const uint8_t* mask, ga_stream_t stream);
/* The same from src into dst: the wrapped env keeps its own (raw) observations,
 * as the inner env of the reference's NormalizedEnv does. */
int ga_obs_normalize_from_f64(int64_t n, int obs_dim, const float* src, float* dst,
                              int64_t ldo, double* mean, double* var, double alpha,
                              const uint8_t* mask, ga_stream_t stream);
int ga_reward_normalize_f64(int64_t n, float* reward, double* mean, double* var,
                            double alpha, double scale, int normalize,
                            ga_stream_t stream);
/* NormalizedEnv.step's action rescale for a Box with finite bounds
 * (envs/normalized_env.py:90-100): out = clip(low + (a + s) * (0.5 (high - low) / s),
 * low, high) in fp32, s = expected_action_scale; low / high: device float[A]. */
int ga_action_rescale_f32(int64_t n, int A, const float* actions, int64_t lda,
                          const float* low, const float* high,
                          float expected_action_scale, float* out, int64_t ldo,
                          ga_stream_t stream);

/* dist.sample() of StochasticPolicy.get_actions
 * (torch/policies/stochastic_policy.py:46-89) + the per-env list appends of
 * VecWorker.step_episode (sampler/vec_worker.py:187-197) for observations and
 * actions. */
typedef struct {
  int64_t n, env_id0;
  int32_t A, kind; /* kind 0 gaussian, 1 categorical */
  const float* head; int64_t ldh;
  const float* log_std; int32_t has_min, has_max; float min_log_std, max_log_std;
  const float* noise; int64_t ldn; /* optional teacher-forced noise */
  uint64_t seed; uint32_t step; int32_t double_softmax;
  const float* obs; int64_t ldo; int32_t obs_dim;
  int64_t col, Tcap;
  float* action; int64_t lda;
  float* obs_buf; float* act_buf; float* head_buf;
} ga_head_args;
int ga_policy_head_sample(const ga_head_args* args, ga_stream_t stream);
/* The same step with the MLP fused in: policy forward (all layers, activations
 * resident in LDS, weights streamed through LDS, hidden layers on MFMA) + the
 * action head + the rollout-buffer writes in ONE launch; `args->head` is
 * ignored.  Supported when every layer input is <= 256 wide and the head <= 32
 * (ga_policy_step_fused_supported); otherwise use ga_mlp_forward_f32 +
 * ga_policy_head_sample. */
int ga_policy_step_fused_supported(const ga_mlp_desc* d);
int ga_policy_step_fused_f32(const ga_mlp_desc* d, const float* params,
                             const ga_head_args* args, ga_stream_t stream);

/* Reward / step-type / episode-end bookkeeping of VecWorker.step_episode and
 * _gather_episode (sampler/vec_worker.py:139-204). */
typedef struct {
  int64_t n, col, Tcap;
  int32_t max_episode_length;
  const float* reward; const uint8_t* step_type; const float* next_obs;
  int64_t ldo; int32_t obs_dim;
  int32_t* ep_t; float* rew_buf; uint8_t* st_buf; uint16_t* tail_buf;
  float* lastobs_buf; uint8_t* done; int32_t* step_eps; int32_t* step_samples;
  int32_t terminal_only; /* 1: FragmentWorker rule, only TERMINAL ends an episode
                            (sampler/fragment_worker.py:114-115) */
} ga_record_args;
int ga_record_step(const ga_record_args* args, ga_stream_t stream);
/* ga_synth_env_step -> ga_record_step -> ga_synth_env_reset(done) of the synthetic
 * environment in one launch (same per-env results; rec->next_obs receives the next
 * observation, or the first observation of the new episode where one ended). */
int ga_synth_env_step_record(const ga_synth_env* env, const ga_record_args* rec,
                             const float* actions, int64_t lda, const float* obs,
                             ga_stream_t stream);
/* ... with NormalizedEnv's observation / reward normalisation fused in (the
 * north star's "fused obs-normalise"): the env steps on its raw observations
 * (raw_obs -> raw_next_obs), the moving statistics are updated and
 * rec->next_obs receives the normalised observation the policy sees next -- for
 * the step's observation (recorded as the terminal one where an episode ends)
 * and again for the first observation of a new episode, in that order, as
 * envs/normalized_env.py:134-151 does.  norm == NULL: no normalisation. */
typedef struct ga_norm_args {
  int32_t normalize_obs, normalize_reward;
  double* obs_mean;      /* [n, obs_dim] float64 moving mean */
  double* obs_var;
  double obs_alpha;
  double* reward_mean;   /* [n] */
  double* reward_var;
  double reward_alpha, reward_scale;
  const float* raw_obs;  /* [n, ldo] the wrapped env's current observations */
  float* raw_next_obs;   /* [n, ldo] where its next observations go */
  /* action rescale (normalized_env.py:90-100) when act_low != NULL: the wrapped env
   * steps on scaled_action (scratch [n, lda]) = ga_action_rescale_f32(policy action) */
  const float* act_low;  /* [act_dim] */
  const float* act_high;
  float expected_action_scale;
  float* scaled_action;
} ga_norm_args;
int ga_synth_env_step_record_norm(const ga_synth_env* env, const ga_record_args* rec,
                                  const ga_norm_args* norm, const float* actions,
                                  int64_t lda, const float* obs, ga_stream_t stream);

/* ga_policy_step_fused_f32 followed, per env and in the same launch, by what
 * ga_synth_env_step_record_norm does (env step with `head->action`, NormalizedEnv
 * statistics, bookkeeping, reset of the finished envs), for n_steps consecutive
 * rollout steps of the synthetic env in ONE launch: a workgroup takes its 32 envs
 * through all of them (envs do not interact within a rollout), alternating between
 * the buffers head->obs / rec->next_obs (norm: raw_obs / raw_next_obs); columns
 * head->col .. head->col + n_steps - 1, Philox counters head->step + s.  Device
 * noise only (head->noise must be null for n_steps > 1). */
int ga_policy_env_step_fused_f32(const ga_mlp_desc* d, const float* params,
                                 const ga_head_args* head, const ga_synth_env* env,
                                 const ga_record_args* rec, const ga_norm_args* norm,
                                 int64_t n_steps, ga_stream_t stream);
/* 1 (default): ga_rollout_synth_steps takes that launch (unless actions are
 * rescaled between policy and env); 0: policy step and env step as two launches. */
int ga_set_fused_env_step(int on);
/* n_steps consecutive vectorised steps (fused policy step, synthetic env step,
 * bookkeeping, reset of finished envs) starting at head->col / head->step,
 * alternating the observation buffers obs_a (current) / obs_b; after an odd
 * number of steps the current observations are in obs_b.  The while-loop body of
 * VecWorker.rollout (sampler/default_worker.py:176-186 + vec_worker.py:176-204)
 * enqueued natively.  With norm != NULL (NormalizedEnv around the synthetic env)
 * obs_a / obs_b hold the normalised observations and raw_a / raw_b, alternating
 * the same way, the env's own. */
int ga_rollout_synth_steps(const ga_mlp_desc* desc, const float* params,
                           const ga_head_args* head, const ga_synth_env* env,
                           const ga_record_args* rec, float* obs_a, float* obs_b,
                           const ga_norm_args* norm, float* raw_a, float* raw_b,
                           int64_t n_steps, ga_stream_t stream);

/* EpisodeBatch.concatenate in completion order (sampler/vec_worker.py:206-219,
 * local_sampler.py:134-166; order = (completion step, env index)). */
int ga_pack_episodes(const uint16_t* tail_buf, int64_t n, int64_t Tcap,
                     int64_t n_steps, const int32_t* ep_base, int32_t* ep_env,
                     int32_t* ep_end, int32_t* ep_len, ga_stream_t stream);
int ga_pack_src_index(const int32_t* ep_env, const int32_t* ep_end,
                      const int32_t* ep_len, const int64_t* ep_off, int64_t n_eps,
                      int64_t Tcap, int32_t* src, ga_stream_t stream);
int ga_gather_rows_f32(const float* src, int64_t ld_src, const int32_t* idx,
                       int64_t rows, int64_t width, float* dst, int64_t ld_dst,
                       ga_stream_t stream);
int ga_gather_f32(const float* src, const int32_t* idx, int64_t n, float* dst,
                  ga_stream_t stream);
int ga_gather_u8(const uint8_t* src, const int32_t* idx, int64_t n, uint8_t* dst,
                 ga_stream_t stream);
/* Minibatch id permutation of BatchDataset
 * (np/optimizers/minibatch_dataset.py:4-35), throughput mode: a keyed Feistel
 * permutation of [0, n) evaluated on the device (the parity mode ships the host
 * np.random.shuffle ids instead). */
int ga_permutation_i32(int64_t n, uint64_t key, int32_t* out, ga_stream_t stream);
/* undiscounted return per episode (log_performance, _functions.py:233-275) */
int ga_episode_sums_f32(const float* rewards, const int64_t* ep_off, int64_t n_eps,
                        double* sums, ga_stream_t stream);

/* ---- one optimisation pass, enqueued natively --------------------------------
 * All minibatches of one epoch of VPG._train (torch/algos/vpg.py:230-293) for
 * one network: forward, fused loss + gradient seed, backward, slab reduction,
 * optional RCCL all-reduce(mean) of the flat gradient, Adam.  Same kernels and
 * order as driving the entry points above one by one; exists because ~14
 * launches per minibatch through ctypes starve the GPU. */
typedef struct {
  const ga_mlp_desc* desc;
  float* params; float* grads; float* exp_avg; float* exp_avg_sq; int64_t n_flat;
  float* acts; float* dacts; float* out; float* dout; int64_t ldo;
  float* slabs; int64_t max_splits;
  int64_t step0;           /* Adam steps already taken */
  double lr, beta1, beta2, eps;
  int32_t learn_std;       /* 0: the log-std slot is not trained */
  const float* X; int64_t ldx; int64_t S;
  const int32_t* perm;     /* S minibatch ids of this pass, NULL = one full batch */
  int64_t mb;
  int32_t kind;            /* 0 Gaussian policy, 1 value (Gaussian NLL),
                              2 categorical policy */
  const float* actions; int64_t lda; const float* old_ll; const float* adv;
  const float* returns;
  int32_t has_min; float min_log_std; int32_t has_max; float max_log_std;
  int32_t algo; float clip; float ent_coeff; int32_t ent_flags;
  float* losses;           /* optional [n_minibatches] per-step losses */
  float* loss_scratch;     /* 1 float, used when losses == NULL */
  double* workspace;
  void* comm; int32_t world; /* RCCL communicator from ga_comm_init_rank, or NULL */
  int32_t double_softmax;    /* kind 2 */
  float grad_scale;          /* with comm: this rank's share S_local / S_global of the
                                minibatch, applied before the all-reduce(sum) */
  int64_t n_mb;              /* > 0 (with perm): the S ids of the pass are split into
                                exactly n_mb minibatches, minibatch k = ids
                                [k*S/n_mb, (k+1)*S/n_mb) -- data parallel ranks hold
                                different S but must issue the same number of
                                all-reduces; 0: ceil(S/mb) minibatches of mb ids,
                                the last one partial (BatchDataset) */
  const float* grad_scales_host; /* optional HOST array [n_mb]: per-minibatch
                                replacement of grad_scale (rows of this rank's
                                minibatch k / rows of the global minibatch k) */
  float* partials;           /* optional scratch for the fused step kernels: with at
                                least ga_update_partials_floats(desc, rows of the
                                largest minibatch) floats the step takes them (last
                                hidden layer + head + loss + gradient seed in one
                                launch, ...; see ga_set_fused_train) */
  int64_t partials_floats;
  int32_t phase;             /* 0: whole steps.  1: gradients only -- forward, loss,
                                backward, slab sum scaled by grad_scale into `grads`;
                                no all-reduce, no optimizer (the Python minibatch
                                loop exchanges and steps itself) */
} ga_update_args;
int ga_update_epoch(const ga_update_args* args, ga_stream_t stream);
/* Host-only (no GPU needed): the split of a pass into minibatches that
 * ga_update_epoch* walks -- ids [*start, *start + *M) of the permutation form
 * minibatch k; returns the number of minibatches of the pass (< 0: bad arguments).
 * S, mb, n_mb as in ga_update_args; has_perm = (perm != NULL).  The Python side's
 * OptimizerWrapper.minibatch_bounds must agree with it on every rank (the reference's
 * BatchDataset split, np/optimizers/minibatch_dataset.py:20-35, and the even split of
 * data-parallel runs); tests/test_host_logic_cpu.py compares the two. */
int64_t ga_minibatch_range(int64_t S, int64_t mb, int64_t n_mb, int has_perm, int64_t k,
                           int64_t* start, int64_t* M);
/* Floats of `partials` scratch the fused step needs for minibatches of up to M rows
 * of this network; 0: the network's shapes take the per-layer kernels (last hidden
 * layer not 64 / 128 / 256 wide, or a head of more than 8 outputs). */
int64_t ga_update_partials_floats(const ga_mlp_desc* desc, int64_t M);
/* 1 (default): ga_update_epoch* steps take the fused kernels when `partials` is
 * given and the shapes allow: the last hidden layer, the head layer, the loss with
 * its gradient seed and the head's weight gradient in ONE launch (the hidden
 * activation never reaches HBM), the data gradient into the first hidden layer and
 * the first layer's weight gradient in one launch, one reduction + Adam launch
 * that also finishes the loss.  Same formulas, other summation orders than the
 * per-layer kernels (0 selects those): results agree to rounding. */
int ga_set_fused_train(int on);
/* 1 (default): networks of two equal tanh hidden layers of 32 or 64 units (<= 32
 * inputs, <= 8 outputs) take forward + loss + backward of a minibatch in ONE launch
 * (every weight in LDS, 64 rows per workgroup) followed by the same reduction +
 * Adam launch; 0: the kernels above.  Needs `partials` like them. */
int ga_set_narrow_step(int on);
/* 1 (default): with two hidden layers and <= 32 inputs (first-layer weights of at
 * most 5120 padded floats) the fused last-hidden-layer kernel computes the first
 * layer's outputs itself, k-chunk by k-chunk, instead of reading them back from a
 * separate launch (they are still written once for the backward pass); 0: the
 * first layer runs as its own launch. */
int ga_set_fused_first_layer(int on);
/* The software-pipelined k-loop of the fused forward kernels (first-layer producer,
 * H1 spill and weight prefetch issued in the shadow of the step's MFMAs; compiled for
 * first layers of 17 .. 20 inputs at 256 units): 1 default, 0 the plain loop (also
 * GARAGE_AMD_PIPELINED_KLOOP=0).  Bit-identical results either way. */
int ga_set_pipelined_kloop(int on);
/* EXPERIMENT, off by default (also GARAGE_AMD_SPLIT_BF16=1): the k-loops of the fused
 * update kernels that have such an instantiation (256-unit networks, first layer in
 * the kernel) run on v_mfma_f32_32x32x16_bf16 with every fp32 operand split exactly
 * into three bf16 terms and the six products i + j <= 2 accumulated in fp32
 * (about 2^-22 relative error per product against 2^-24 of the exact fp32 MFMA).
 * Results differ from the default in the last bits; the default stays exact fp32. */
int ga_set_split_bf16(int on);
/* The policy pass and the value-function pass of one epoch, minibatch by
 * minibatch alternately on two streams.  The reference runs them back to back
 * (vpg.py:244-248); they share no written state, so the results are identical
 * and the two launch chains overlap on the device.  The two argument sets must
 * not share parameter, slab, activation or reduction buffers. */
int ga_update_epoch_pair(const ga_update_args* a, ga_stream_t stream_a,
                         const ga_update_args* b, ga_stream_t stream_b);

/* RCCL communicator for the data-parallel gradient all-reduce (new: the
 * reference has no collective on this path).
 * ga_comm_unique_id fills 128 bytes on rank 0 (HOST pointer) to be broadcast by
 * the caller; ga_comm_init_rank returns an opaque handle. */
/* The all-reduce ga_update_epoch* calls on args->comm between the slab reduction
 * and Adam: fn(comm, buf, n, stream) sums buf over the ranks in place, returns 0.
 * ga_comm_init_rank installs RCCL's; a caller with another transport (or a test
 * standing in for the second rank) installs its own. */
typedef int (*ga_allreduce_fn)(void* comm, float* buf, int64_t n, void* stream);
void ga_set_allreduce_hook(ga_allreduce_fn fn);
int ga_comm_available(void); /* 1 if librccl could be loaded in this process */
int ga_comm_unique_id(void* id128_host);
void* ga_comm_init_rank(const void* id128_host, int rank, int world);
int ga_comm_allreduce_sum_f32(void* comm, float* buf, int64_t n, ga_stream_t stream);
/* ranks of the communicator as RCCL reports them (ncclCommCount); < 0 on error */
int ga_comm_count(void* comm);
int ga_comm_destroy(void* comm);
/* ga_update_epoch_pair with communicators: the two networks' all-reduces sit on two
 * streams and two communicators.  1 (default): every all-reduce waits (HIP event)
 * for the previously enqueued one of the OTHER network, so each GPU executes them
 * in the host's issue order, identical on every rank -- no rank can sit in network
 * A's collective while its peer sits in network B's.  0: no such edges (the two
 * collectives may run concurrently; they are small enough to co-reside). */
int ga_set_ordered_allreduce(int on);
/* ga_update_epoch_pair, one process: 1 (opt-in, also GARAGE_AMD_MERGED_PAIR=1) step k of
 * BOTH passes as four launches on stream_a, each a grid over the tiles of both networks
 * (when both take the fused 256-wide kernels with the same shapes; stream_b is ordered
 * around the epoch); 0 (default) two four-launch chains on the two streams.
 * Bit-identical results; the merged schedule is the same every time but measured
 * slower. */
int ga_set_merged_pair(int on);

/* ---- measurement ----------------------------------------------------------
 * Optional HIP-event timing of every GEMM / scan launch on its own stream
 * (bench.py's roofline leg; no reference counterpart).  kinds: 0..2 =
 * gemm_f32_kernel<128,128,..> forward / data-grad / weight-grad, 3..5 = the
 * <256,32,..> variants, 6 = gae_scan_kernel.  ga_prof_collect synchronises and
 * fills out_host[kind*3 + {0,1,2}] = {total ms, total flops or bytes, launches}
 * (a HOST pointer). */
int ga_prof_enable(int on);
int ga_prof_collect(double* out_host, int n_kinds);
/* Launches of kernel kind `kind` (csrc/prof.h: 9 = fwd_head_loss_kernel, 10 =
 * dgrad_wgrad0_kernel, 11 = narrow_train_kernel, 12 = mlp_eval_forward_kernel, 13 =
 * a whole rollout in one policy_step_fused_kernel launch, ...)
 * since the library was loaded, counted whether timing is on or not; -1 for an
 * unknown kind.  Tests use it to assert WHICH kernels an update dispatched to. */
int64_t ga_launch_count(int kind);

#ifdef __cplusplus
}
#endif
#endif /* GARAGE_AMD_H_ */
