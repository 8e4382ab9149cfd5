// Arguments of the one-launch small-minibatch optimizer step (small_step.hip),
// filled by the epoch loop (update.cpp).  Not part of the C ABI.
#pragma once
#include <stdint.h>

struct ga_small_step_args {
  float* params; float* exp_avg; float* exp_avg_sq;
  int64_t w_off[3], b_off[3];
  int in_w, H, out_w, M;
  const float* X; int64_t ldx; const int32_t* idx;
  int kind; int double_softmax;
  const float* actions; int64_t lda; const float* old_ll; const float* adv;
  const float* returns;
  int algo; float clip;
  int has_min, has_max; float min_log_std, max_log_std;
  float ent_coeff; int ent_flags;
  int64_t step; double lr, beta1, beta2, eps;
  int learn_std;
  float* xh2; float* xdz;  // [64][H] floats each
  unsigned* bar;           // {arrival count, phase word}: 0 between launches
  float* loss_out;
  int* fault;              // raised when a launch aborted (a grid barrier gave up)
};

extern "C" int ga_small_step_supported(int n_layers, const int* dims, int64_t M);
// `concurrent` such grids (the policy and the value chain run side by side) fit on
// the device at once, by the occupancy query: the grid barriers need every
// workgroup resident
extern "C" int ga_small_step_resident(int H, int concurrent);
extern "C" int ga_small_step(const ga_small_step_args* a, void* stream);
