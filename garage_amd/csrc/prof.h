// Launch-timing hooks (see prof.cpp).  Kinds index the kernels bench.py reports.
#pragma once
#include <hip/hip_runtime.h>

enum GaProfKind {
  GA_PROF_GEMM_NT_128 = 0,  // gemm_f32_kernel<128,128,2,4,true,true>   forward
  GA_PROF_GEMM_NN_128 = 1,  // gemm_f32_kernel<128,128,2,4,true,false>  data grad
  GA_PROF_GEMM_TN_128 = 2,  // gemm_f32_kernel<128,128,2,4,false,false> weight grad
  GA_PROF_GEMM_NT_256 = 3,  // gemm_f32_kernel<128,32,4,1,true,true>
  GA_PROF_GEMM_NN_256 = 4,  // gemm_f32_kernel<128,32,4,1,true,false>
  GA_PROF_GEMM_TN_256 = 5,  // gemm_f32_kernel<128,32,4,1,false,false>
  GA_PROF_GAE_SCAN = 6,     // gae_scan_kernel<*>
  GA_PROF_SKINNY_FWD = 7,   // skinny_fwd_kernel<*>   (work = algorithmic bytes)
  GA_PROF_SKINNY_WGRAD = 8, // skinny_wgrad_kernel<*> (work = algorithmic bytes)
  GA_PROF_FUSED_FWD = 9,    // fwd_head_loss_kernel<*>  (last hidden layer + head + loss)
  GA_PROF_FUSED_DGRAD = 10, // dgrad_wgrad0_kernel<*>   (data grad + first-layer wgrad)
  GA_PROF_NARROW_STEP = 11, // narrow_train_kernel<*>  (whole forward + backward, H <= 64)
  GA_PROF_EVAL_FWD = 12,    // mlp_eval_forward_kernel<*> (whole MLP, outputs only)
  GA_PROF_ROLLOUT = 13,     // policy_step_fused_kernel<true>: a whole rollout (policy +
                            // env + bookkeeping for n_steps > 1) in one launch; counted
                            // only (ga_launch_count), never timed
  GA_PROF_KINDS = 14
};

// When profiling is on, hands out a (start, stop) event pair to attach to ONE
// kernel dispatch with hipExtLaunchKernelGGL: the elapsed time is then the
// kernel's own duration (what rocprofv3 reports), with no launch gaps in it.
// When off, both stay null and the launch is an ordinary one.
void ga_prof_events(int kind, double work, hipEvent_t* start, hipEvent_t* stop);
// Counts a launch of `kind` without timing it (ga_launch_count).
void ga_prof_count(int kind);
