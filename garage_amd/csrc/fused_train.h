// Entry points of the fused optimizer-step kernels (fused_train.hip) and of the MLP
// layer ranges they combine with (gemm.hip), used by the epoch loop (update.cpp).
// Not part of the C ABI: ga_update_epoch* is what callers see.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct ga_fused_loss_args {
  int kind;                  // 0 Gaussian policy, 1 value NLL, 2 categorical policy
  const float* actions; int64_t lda;
  const float* old_ll; const float* adv; const float* returns;
  const int32_t* idx;        // minibatch row m is sample idx[m] (null: m)
  const float* log_std; int has_min, has_max; float min_log_std, max_log_std;
  int A;                     // head width
  int algo; float clip; float ent_coeff; int ent_flags;
  int double_softmax;
};

struct ga_fused_region {
  int64_t beg, n;            // flat parameter range [beg, beg + n)
  const float* src;          // partial 0 of element 0
  int64_t stride;            // floats between partials
  int n_part;
};

extern "C" {
int ga_fused_width_ok(int width);       // 64, 128 or 256 units
int64_t ga_fused_tiles(int64_t M);      // workgroups (= partial sets) for M rows
// The layer under the last hidden one when that is the network's FIRST layer and
// the kernel is to produce its outputs itself (H = tanh(X W^T + b), X gathered
// through loss->idx) instead of reading them: in_w <= 32, K * round4(in_w) <=
// ga_fused_first_layer_ok's bound.  H is written too (the backward pass reads it).
typedef struct ga_fused_first_layer {
  const float* X;
  int64_t ldx;
  const float* W;  // [K][round4(in_w)]
  const float* b;  // [K]
  int in_w;
  float* H;        // [M][ldh]
  int64_t ldh;
} ga_fused_first_layer;
int ga_fused_first_layer_ok(int in_w, int K);
// last hidden layer + head + loss + gradient seed + head weight-gradient shares;
// hpart: [tiles][8 * width + 8] floats, lpart: [tiles][2] doubles.  first != null:
// A / lda / a_idx are ignored, the operand comes from `first`
int ga_fused_fwd_head_loss(const float* A, int64_t lda, const int32_t* a_idx,
                           const float* W, int64_t ldw, const float* bias, int64_t M,
                           int width, int K, const float* head_W, int64_t head_ldw,
                           const float* head_bias, const ga_fused_loss_args* loss,
                           float* dZ, int64_t lddz, float* hpart, double* lpart,
                           const ga_fused_first_layer* first, hipStream_t stream);
// The whole MLP (two hidden tanh layers of the shapes above, <= 8 linear outputs) for
// its OUTPUTS only: out[m] = MLP(X[idx ? idx[m] : m]); no activation reaches memory.
int ga_fused_eval_supported(int n_layers, const int* dims);
int ga_fused_eval_forward(const float* X, int64_t ldx, const int32_t* idx, int64_t M,
                          const int* dims, const float* W1, const float* b1,
                          const float* W2, const float* b2, const float* Wh,
                          const float* bh, float* out, int64_t ldo, hipStream_t stream);
// data gradient into the first hidden layer + first-layer weight / bias gradient
// shares; wpart: [tiles][width * round4(in_w) + width] floats
int ga_fused_dgrad_wgrad0(const float* dZ2, int64_t lddz, const float* W2, int64_t ldw,
                          int64_t M, int width, int K, const float* H1, int64_t ldh,
                          const float* X, int64_t ldx, const int32_t* idx, int in_w,
                          float* wpart, hipStream_t stream);
// narrow_step.hip: forward + loss + backward of a 2 x H network (H = 32 or 64) in one
// launch; part: [tiles][ga_narrow_step_stride] floats, lpart: [tiles][2] doubles
int ga_narrow_step_supported(int n_layers, const int* dims);
int64_t ga_narrow_step_stride(int in_w, int H);
int ga_narrow_train_step(const float* params, const int64_t* w_off, const int64_t* b_off,
                         int in_w, int H, int out_w, const float* X, int64_t ldx,
                         int64_t M, const ga_fused_loss_args* loss, float* part,
                         double* lpart, hipStream_t stream);
// ---- two networks in one launch each (the policy's and the value function's step k):
// same shapes for both (width 256, first layer in the kernel), same row count
int ga_fused_pair_supported(int width, int K, int in_w);
int ga_fused_fwd_head_loss_pair(
    int64_t M, int width, int K,
    const float* Wa, int64_t ldwa, const float* biasa, const float* head_Wa,
    int64_t head_ldwa, const float* head_biasa, const ga_fused_loss_args* lossa,
    float* dZa, int64_t lddza, float* hparta, double* lparta,
    const ga_fused_first_layer* firsta,
    const float* Wb, int64_t ldwb, const float* biasb, const float* head_Wb,
    int64_t head_ldwb, const float* head_biasb, const ga_fused_loss_args* lossb,
    float* dZb, int64_t lddzb, float* hpartb, double* lpartb,
    const ga_fused_first_layer* firstb, hipStream_t stream);
int ga_fused_dgrad_wgrad0_pair(
    int64_t M, int width, int K, int in_w,
    const float* dZ2a, int64_t lddza, const float* W2a, int64_t ldwa, const float* H1a,
    int64_t ldha, const float* Xa, int64_t ldxa, const int32_t* idxa, float* wparta,
    const float* dZ2b, int64_t lddzb, const float* W2b, int64_t ldwb, const float* H1b,
    int64_t ldhb, const float* Xb, int64_t ldxb, const int32_t* idxb, float* wpartb,
    hipStream_t stream);
// gemm.hip: the weight-gradient GEMM of the middle layer (dW2 = dZ2^T H1, split-K
// slabs + bias column sums) of two 3-layer networks in one grid
int ga_wgrad_mid_pair(int64_t M, int64_t n_splits, int out_w, int in_w,
                      const float* dza, const float* ina, float* slabs_wa, float* slabs_ba,
                      int64_t slab_stride_a,
                      const float* dzb, const float* inb, float* slabs_wb, float* slabs_bb,
                      int64_t slab_stride_b, hipStream_t stream);
typedef struct ga_reduce_net {
  const ga_fused_region* regions; int n_regions;
  float* params; float* grads; float* exp_avg; float* exp_avg_sq;
  int64_t step; double lr, beta1, beta2, eps; float scale; int do_adam, zero_slot0;
  const double* lpart; int n_lpart; int64_t M; const ga_fused_loss_args* loss;
  float* loss_out;
} ga_reduce_net;
int ga_reduce_regions_adam_pair(const ga_reduce_net* a, const ga_reduce_net* b,
                                hipStream_t stream);
int ga_reduce_regions_adam(const ga_fused_region* regions, int n_regions, float* params,
                           float* grads, float* exp_avg, float* exp_avg_sq, int64_t step,
                           double lr, double beta1, double beta2, double eps, float scale,
                           int do_adam, int zero_slot0, const double* lpart, int n_lpart,
                           int64_t M, const ga_fused_loss_args* loss, float* loss_out,
                           hipStream_t stream);
/* 1 while the opt-in split-operand (3 x bf16) k-loops are selected
 * (ga_set_split_bf16 / GARAGE_AMD_SPLIT_BF16=1). */
int ga_split_bf16_enabled(void); /* ... for the weight-gradient GEMM */
int ga_split_bf16_any(void);     /* ... for any kernel */
int ga_split_bf16_gemm(void);    /* ... for the per-layer forward / data-gradient GEMMs */
/* W [rows][ld] (cols valid) as the B operand of a split-operand k-loop: three bf16
 * planes in fragment order (fused_train.hip: split_planes_kernel), both dimensions
 * padded to multiples of 32; bwd = 0: B(k, n) = W[n][k], 1: B(k, n) = W[k][n].
 * Recomputed by every call, on `stream`; plane pl at + pl * round32(rows) *
 * round32(cols). */
const uint16_t* ga_weight_planes(const float* W, int64_t ld, int rows, int cols, int bwd,
                                 hipStream_t stream);
/* The next ga_reduce_regions_adam of this thread also rewrites the planes of the
 * [rows][cols] weight matrix at flat index flat_beg (the step's forward launch used
 * them); ga_planes_epoch_begin: planes written that way are trusted only until the
 * epoch call that wrote them returns. */
void ga_reduce_planes_hint(int64_t flat_beg, int rows, int cols);
void ga_planes_epoch_begin(void);
}
