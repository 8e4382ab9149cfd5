// One optimisation epoch (all minibatches of one pass) enqueued from C++.
//
// VPG._train (torch/algos/vpg.py:230-248) iterates minibatches in Python and
// every minibatch costs ~14 kernel launches here; driven from Python through
// ctypes that is ~300 us of host time per minibatch and the GPU starves.  This
// entry point walks the minibatches of one pass in native code (same kernels,
// same order: forward, fused loss + gradient seed, backward, slab reduction,
// [RCCL all-reduce], Adam), so the host cost is the HIP launch itself.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/garage_amd.h"

void ga_set_error(const char* fmt, ...);

typedef int (*ga_allreduce_fn)(void* comm, float* buf, int64_t n, void* stream);
static ga_allreduce_fn g_allreduce = nullptr;

extern "C" void ga_set_allreduce_hook(ga_allreduce_fn fn) { g_allreduce = fn; }

extern "C" int ga_update_epoch(const ga_update_args* a, ga_stream_t stream) {
  if (!a || !a->desc || !a->params || !a->X || !a->workspace) {
    ga_set_error("ga_update_epoch: null pointer");
    return -1;
  }
  if (a->S <= 0 || (a->perm && a->mb <= 0)) {
    ga_set_error("ga_update_epoch: bad sizes");
    return -1;
  }
  const int64_t mb = a->perm ? a->mb : a->S;
  const int64_t n_mb = (a->S + mb - 1) / mb;
  const int L = a->desc->n_layers;
  const int out_w = a->desc->dims[L];
  for (int64_t k = 0; k < n_mb; ++k) {
    const int64_t M = (k == n_mb - 1) ? (a->S - k * mb) : mb;
    const int32_t* idx = a->perm ? a->perm + k * mb : nullptr;
    int rc = ga_mlp_forward_f32(a->desc, a->params, a->X, a->ldx, idx, M, a->acts,
                                a->out, a->ldo, stream);
    if (rc) return rc;
    const int64_t splits = ga_mlp_backward_splits(a->desc, M);
    if (splits > a->max_splits) {
      ga_set_error("ga_update_epoch: slab workspace too small");
      return -1;
    }
    float* loss_slot = a->losses ? a->losses + k : a->loss_scratch;
    if (a->kind == 0) {
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
    rc = ga_mlp_backward_f32(a->desc, a->params, a->X, a->ldx, idx, M, a->acts,
                             a->dout, a->ldo, a->dacts, a->slabs, a->n_flat, splits,
                             stream);
    if (rc) return rc;
    const float scale = (a->comm && a->world > 1) ? 1.0f / (float)a->world : 1.0f;
    rc = ga_reduce_slabs_f32(a->slabs, splits, a->n_flat, a->n_flat, scale, a->grads,
                             stream);
    if (rc) return rc;
    if (!a->learn_std) {
      if (hipMemsetAsync(a->grads, 0, sizeof(float), (hipStream_t)stream) !=
          hipSuccess) {
        ga_set_error("ga_update_epoch: memset failed");
        return -2;
      }
    }
    if (a->comm && a->world > 1) {
      if (!g_allreduce) {
        ga_set_error("ga_update_epoch: no all-reduce hook installed");
        return -1;
      }
      rc = g_allreduce(a->comm, a->grads, a->n_flat, stream);
      if (rc) {
        ga_set_error("ga_update_epoch: all-reduce failed (%d)", rc);
        return -2;
      }
    }
    rc = ga_adam_step_f32(a->params, a->grads, a->exp_avg, a->exp_avg_sq, a->n_flat,
                          a->step0 + k + 1, a->lr, a->beta1, a->beta2, a->eps, stream);
    if (rc) return rc;
  }
  return 0;
}
