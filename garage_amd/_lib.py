"""ctypes binding of ``libgarage_amd.so`` (the C ABI in ``include/garage_amd.h``).

There is no CPU fallback: if the library is missing or a symbol is absent the
import of any compute path fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, '_C', 'libgarage_amd.so')

c_i32, c_i64, c_u32, c_u64 = C.c_int32, C.c_int64, C.c_uint32, C.c_uint64
c_f32, c_f64, c_int, ptr = C.c_float, C.c_double, C.c_int, C.c_void_p


class MlpDesc(C.Structure):
    """``ga_mlp_desc``."""
    _fields_ = [('n_layers', c_i32), ('dims', c_i32 * 9), ('w_off', c_i64 * 8),
                ('b_off', c_i64 * 8), ('act_off', c_i64 * 8),
                ('hidden_act', c_i32), ('output_act', c_i32),
                ('layer_norm', c_i32), ('pad_', c_i32), ('ln_off', c_i64 * 8),
                ('lnx_off', c_i64 * 8), ('lns_off', c_i64 * 8)]


class SynthEnv(C.Structure):
    """``ga_synth_env``."""
    _fields_ = [('n', c_i64), ('env_id0', c_i64), ('obs_dim', c_i32),
                ('act_dim', c_i32), ('discrete', c_i32), ('min_len', c_i32),
                ('max_len', c_i32), ('seed', c_u64), ('episode', ptr),
                ('t', ptr), ('len', ptr)]


class HeadArgs(C.Structure):
    """``ga_head_args``."""
    _fields_ = [('n', c_i64), ('env_id0', c_i64), ('A', c_i32), ('kind', c_i32),
                ('head', ptr), ('ldh', c_i64), ('log_std', ptr),
                ('has_min', c_i32), ('has_max', c_i32), ('min_log_std', c_f32),
                ('max_log_std', c_f32), ('noise', ptr), ('ldn', c_i64),
                ('seed', c_u64), ('step', c_u32), ('double_softmax', c_i32),
                ('obs', ptr), ('ldo', c_i64), ('obs_dim', c_i32),
                ('col', c_i64), ('Tcap', c_i64), ('action', ptr),
                ('lda', c_i64), ('obs_buf', ptr), ('act_buf', ptr),
                ('head_buf', ptr)]


class RecordArgs(C.Structure):
    """``ga_record_args``."""
    _fields_ = [('n', c_i64), ('col', c_i64), ('Tcap', c_i64),
                ('max_episode_length', c_i32), ('reward', ptr),
                ('step_type', ptr), ('next_obs', ptr), ('ldo', c_i64),
                ('obs_dim', c_i32), ('ep_t', ptr), ('rew_buf', ptr),
                ('st_buf', ptr), ('tail_buf', ptr), ('lastobs_buf', ptr),
                ('done', ptr), ('step_eps', ptr), ('step_samples', ptr),
                ('terminal_only', c_i32)]


class NormArgs(C.Structure):
    """``ga_norm_args``."""
    _fields_ = [('normalize_obs', c_i32), ('normalize_reward', c_i32),
                ('obs_mean', ptr), ('obs_var', ptr), ('obs_alpha', c_f64),
                ('reward_mean', ptr), ('reward_var', ptr),
                ('reward_alpha', c_f64), ('reward_scale', c_f64),
                ('raw_obs', ptr), ('raw_next_obs', ptr), ('act_low', ptr),
                ('act_high', ptr), ('expected_action_scale', c_f32),
                ('scaled_action', ptr)]


class UpdateArgs(C.Structure):
    """``ga_update_args``."""
    _fields_ = [('desc', C.POINTER(MlpDesc)), ('params', ptr), ('grads', ptr),
                ('exp_avg', ptr), ('exp_avg_sq', ptr), ('n_flat', c_i64),
                ('acts', ptr), ('dacts', ptr), ('out', ptr), ('dout', ptr),
                ('ldo', c_i64), ('slabs', ptr), ('max_splits', c_i64),
                ('step0', c_i64), ('lr', c_f64), ('beta1', c_f64),
                ('beta2', c_f64), ('eps', c_f64), ('learn_std', c_i32),
                ('X', ptr), ('ldx', c_i64), ('S', c_i64), ('perm', ptr),
                ('mb', c_i64), ('kind', c_i32), ('actions', ptr),
                ('lda', c_i64), ('old_ll', ptr), ('adv', ptr),
                ('returns', ptr), ('has_min', c_i32), ('min_log_std', c_f32),
                ('has_max', c_i32), ('max_log_std', c_f32), ('algo', c_i32),
                ('clip', c_f32), ('ent_coeff', c_f32), ('ent_flags', c_i32),
                ('losses', ptr), ('loss_scratch', ptr), ('workspace', ptr),
                ('comm', ptr), ('world', c_i32), ('double_softmax', c_i32),
                ('grad_scale', c_f32), ('n_mb', c_i64),
                ('grad_scales_host', ptr), ('partials', ptr),
                ('partials_floats', c_i64), ('phase', c_i32)]


ABI_VERSION = 4  # ga_abi_version() of the library these structs mirror

# name -> (restype, argtypes); mirrors include/garage_amd.h one to one.
SIGNATURES = {
    'ga_abi_version': (c_int, []),
    'ga_last_error': (C.c_char_p, []),
    'ga_gae_scan_f32': (c_int, [ptr, ptr, ptr, ptr, ptr, c_i64, c_i64, c_i64,
                                c_i64, c_int, c_int, c_f64, c_f64, c_f32,
                                c_f32, ptr, ptr, ptr]),
    'ga_act_slope_mul_f32': (c_int, [ptr, c_i64, ptr, c_i64, c_i64, c_int, c_int,
                                     ptr]),
    'ga_set_gae_fixed_fast_path': (c_int, [c_int]),
    'ga_set_gae_rows_steps_per_lane': (c_int, [c_int]),
    'ga_mlp_forward_f32': (c_int, [C.POINTER(MlpDesc), ptr, ptr, c_i64, ptr,
                                   c_i64, ptr, ptr, c_i64, ptr]),
    'ga_mlp_forward_fused_f32': (c_int, [C.POINTER(MlpDesc), ptr, ptr, c_i64,
                                         ptr, c_i64, ptr, ptr, c_i64, ptr]),
    'ga_set_fused_forward': (c_int, [c_int]),
    'ga_mlp_forward_eval_supported': (c_int, [C.POINTER(MlpDesc)]),
    'ga_set_eval_forward': (c_int, [c_int]),
    'ga_set_skinny_kernels': (c_int, [c_int]),
    'ga_set_fused_head_dgrad': (c_int, [c_int]),
    'ga_set_fused_head_forward': (c_int, [c_int]),
    'ga_set_small_m_gemm': (c_int, [c_int]),
    'ga_set_small_step': (c_int, [c_int]),
    'ga_set_small_step_resident_cap': (c_int, [c_int]),
    'ga_set_small_step_max_polls': (c_int, [c_int]),
    'ga_small_step_launches': (c_i64, []),
    'ga_small_step_debug': (c_int, [ptr]),
    'ga_narrow_step_debug': (c_int, [ptr]),
    'ga_policy_step_debug': (c_int, [ptr]),
    'ga_fused_fwd_debug': (c_int, [ptr]),
    'ga_fused_fwd_debug_skew': (c_int, [ptr, c_int]),
    'ga_fused_dgrad_debug': (c_int, [ptr, c_int, c_int]),
    'ga_set_one_launch_losses': (c_int, [c_int]),
    'ga_mlp_backward_splits': (c_i64, [C.POINTER(MlpDesc), c_i64]),
    'ga_mlp_backward_f32': (c_int, [C.POINTER(MlpDesc), ptr, ptr, c_i64, ptr,
                                    c_i64, ptr, ptr, c_i64, ptr, ptr, c_i64,
                                    c_i64, ptr]),
    'ga_mlp_jvp_f32': (c_int, [C.POINTER(MlpDesc), ptr, ptr, ptr, c_i64, ptr,
                               c_i64, ptr, ptr, ptr, c_i64, ptr]),
    'ga_dot_f32': (c_int, [ptr, ptr, c_i64, ptr, ptr]),
    'ga_axpby_f32': (c_int, [c_f64, ptr, c_f64, ptr, c_i64, ptr]),
    'ga_fisher_seed_gaussian_f32': (c_int, [ptr, c_i64, c_i64, c_int, ptr, c_int,
                                            c_f32, c_int, c_f32, ptr, c_i64,
                                            ptr]),
    'ga_fisher_seed_categorical_f32': (c_int, [ptr, c_i64, ptr, c_i64, c_i64,
                                               c_int, c_int, ptr, c_i64, ptr]),
    'ga_gemm_nt_f32': (c_int, [ptr, c_i64, ptr, c_i64, ptr, c_i64, c_i64,
                               c_i64, c_i64, ptr]),
    'ga_reduction_workspace_doubles': (c_i64, []),
    'ga_ppo_gaussian_loss_f32': (c_int, [ptr, c_i64, ptr, c_i64, ptr, ptr, ptr,
                                         ptr, c_int, c_f32, c_int, c_f32,
                                         c_i64, c_int, c_int, c_f32, c_f32,
                                         c_int, ptr, ptr, ptr, ptr, c_i64,
                                         c_i64, ptr, ptr]),
    'ga_ppo_categorical_loss_f32': (c_int, [ptr, c_i64, ptr, c_i64, ptr, ptr,
                                            ptr, c_i64, c_int, c_int, c_int,
                                            c_f32, c_f32, c_int, ptr, ptr, ptr,
                                            ptr, ptr, ptr, c_i64, c_i64, ptr,
                                            ptr]),
    'ga_categorical_kl_f32': (c_int, [ptr, ptr, c_i64, c_i64, c_int, c_int,
                                      ptr, ptr, ptr]),
    'ga_gaussian_nll_loss_f32': (c_int, [ptr, c_i64, ptr, ptr, ptr, c_i64, ptr,
                                         ptr, ptr, c_i64, c_i64, ptr, ptr]),
    'ga_head_loss_supported': (c_int, [c_int, c_int]),
    'ga_set_fused_head_loss': (c_int, [c_int]),
    'ga_head_ppo_gaussian_loss_f32': (c_int, [
        ptr, c_i64, ptr, c_i64, ptr, c_int, ptr, c_i64, ptr, c_i64, ptr, ptr,
        ptr, ptr, c_int, c_f32, c_int, c_f32, c_i64, c_int, c_int, c_f32, c_f32,
        c_int, ptr, c_i64, ptr, ptr, ptr, c_i64, c_i64, ptr, ptr]),
    'ga_head_gaussian_nll_loss_f32': (c_int, [
        ptr, c_i64, ptr, ptr, c_int, ptr, c_i64, ptr, ptr, ptr, c_i64, ptr,
        c_i64, ptr, ptr, c_i64, c_i64, ptr, ptr]),
    'ga_gaussian_kl_f32': (c_int, [ptr, ptr, c_i64, c_i64, c_int, c_f32, c_f32,
                                   ptr, ptr, ptr]),
    'ga_reduce_slabs_f32': (c_int, [ptr, c_i64, c_i64, c_i64, c_f32, ptr, ptr]),
    'ga_reduce_adam_f32': (c_int, [ptr, c_i64, c_i64, ptr, ptr, ptr, ptr, c_i64,
                                   c_i64, c_f64, c_f64, c_f64, c_f64, c_int,
                                   ptr]),
    'ga_optimizer_step_f32': (c_int, [c_int, ptr, ptr, ptr, ptr, ptr, c_i64,
                                      c_i64, C.POINTER(c_f64), c_int, ptr]),
    'ga_adam_step_f32': (c_int, [ptr, ptr, ptr, ptr, c_i64, c_i64, c_f64,
                                 c_f64, c_f64, c_f64, ptr]),
    'ga_stats_f32': (c_int, [ptr, c_i64, c_int, ptr, ptr, ptr]),
    'ga_adv_center_f32': (c_int, [ptr, c_i64, ptr, c_f32, ptr]),
    'ga_sub_scalar_f32': (c_int, [ptr, c_i64, ptr, ptr]),
    'ga_synth_env_reset': (c_int, [C.POINTER(SynthEnv), ptr, ptr, c_i64, ptr]),
    'ga_synth_env_step': (c_int, [C.POINTER(SynthEnv), ptr, c_i64, ptr, ptr,
                                  c_i64, ptr, ptr, ptr]),
    'ga_obs_normalize_f64': (c_int, [c_i64, c_int, ptr, c_i64, ptr, ptr, c_f64,
                                     ptr, ptr]),
    'ga_obs_normalize_from_f64': (c_int, [c_i64, c_int, ptr, ptr, c_i64, ptr,
                                          ptr, c_f64, ptr, ptr]),
    'ga_reward_normalize_f64': (c_int, [c_i64, ptr, ptr, ptr, c_f64, c_f64,
                                        c_int, ptr]),
    'ga_action_rescale_f32': (c_int, [c_i64, c_int, ptr, c_i64, ptr, ptr, c_f32,
                                      ptr, c_i64, ptr]),
    'ga_policy_head_sample': (c_int, [C.POINTER(HeadArgs), ptr]),
    'ga_policy_step_fused_supported': (c_int, [C.POINTER(MlpDesc)]),
    'ga_policy_step_fused_f32': (c_int, [C.POINTER(MlpDesc), ptr,
                                         C.POINTER(HeadArgs), ptr]),
    'ga_record_step': (c_int, [C.POINTER(RecordArgs), ptr]),
    'ga_synth_env_step_record': (c_int, [C.POINTER(SynthEnv),
                                         C.POINTER(RecordArgs), ptr, c_i64, ptr,
                                         ptr]),
    'ga_synth_env_step_record_norm': (c_int, [C.POINTER(SynthEnv),
                                              C.POINTER(RecordArgs),
                                              C.POINTER(NormArgs), ptr, c_i64,
                                              ptr, ptr]),
    'ga_policy_env_step_fused_f32': (c_int, [C.POINTER(MlpDesc), ptr,
                                             C.POINTER(HeadArgs),
                                             C.POINTER(SynthEnv),
                                             C.POINTER(RecordArgs),
                                             C.POINTER(NormArgs), c_i64, ptr]),
    'ga_set_fused_env_step': (c_int, [c_int]),
    'ga_rollout_synth_steps': (c_int, [C.POINTER(MlpDesc), ptr,
                                       C.POINTER(HeadArgs),
                                       C.POINTER(SynthEnv),
                                       C.POINTER(RecordArgs), ptr, ptr,
                                       C.POINTER(NormArgs), ptr, ptr, c_i64,
                                       ptr]),
    'ga_pack_episodes': (c_int, [ptr, c_i64, c_i64, c_i64, ptr, ptr, ptr, ptr,
                                 ptr]),
    'ga_pack_src_index': (c_int, [ptr, ptr, ptr, ptr, c_i64, c_i64, ptr, ptr]),
    'ga_gather_rows_f32': (c_int, [ptr, c_i64, ptr, c_i64, c_i64, ptr, c_i64,
                                   ptr]),
    'ga_gather_f32': (c_int, [ptr, ptr, c_i64, ptr, ptr]),
    'ga_gather_u8': (c_int, [ptr, ptr, c_i64, ptr, ptr]),
    'ga_permutation_i32': (c_int, [c_i64, c_u64, ptr, ptr]),
    'ga_episode_sums_f32': (c_int, [ptr, ptr, c_i64, ptr, ptr]),
    'ga_minibatch_range': (c_i64, [c_i64, c_i64, c_i64, c_int, c_i64,
                                   C.POINTER(c_i64), C.POINTER(c_i64)]),
    'ga_update_epoch': (c_int, [C.POINTER(UpdateArgs), ptr]),
    'ga_update_partials_floats': (c_i64, [C.POINTER(MlpDesc), c_i64]),
    'ga_set_fused_train': (c_int, [c_int]),
    'ga_set_narrow_step': (c_int, [c_int]),
    'ga_set_fused_first_layer': (c_int, [c_int]),
    'ga_set_pipelined_kloop': (c_int, [c_int]),
    'ga_set_split_bf16': (c_int, [c_int]),
    'ga_update_epoch_pair': (c_int, [C.POINTER(UpdateArgs), ptr,
                                     C.POINTER(UpdateArgs), ptr]),
    'ga_set_allreduce_hook': (None, [ptr]),
    'ga_comm_available': (c_int, []),
    'ga_comm_unique_id': (c_int, [ptr]),
    'ga_comm_init_rank': (ptr, [ptr, c_int, c_int]),
    'ga_comm_allreduce_sum_f32': (c_int, [ptr, ptr, c_i64, ptr]),
    'ga_comm_count': (c_int, [ptr]),
    'ga_comm_destroy': (c_int, [ptr]),
    'ga_set_ordered_allreduce': (c_int, [c_int]),
    'ga_set_merged_pair': (c_int, [c_int]),
    'ga_prof_enable': (c_int, [c_int]),
    'ga_prof_collect': (c_int, [C.POINTER(c_f64), c_int]),
    'ga_launch_count': (c_i64, [c_int]),
}


class GarageAmdError(RuntimeError):
    """A C-ABI call returned a negative status."""


_lib = None


def load():
    """Load the shared library (once) and attach the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'garage_amd: HIP extension not built: {} is missing. Run `make` '
            '(or `python -c "import __graft_entry__ as g; g.build()"`) in the '
            'repository root. There is no CPU fallback.'.format(LIB_PATH))
    lib = C.CDLL(LIB_PATH)
    lib.ga_abi_version.restype = c_int
    if lib.ga_abi_version() != ABI_VERSION:
        raise ImportError(
            'garage_amd: {} was built from another version of the sources (ABI {} '
            'against {} here: the argument structs differ). Run `make`.'.format(
                LIB_PATH, lib.ga_abi_version(), ABI_VERSION))
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError = missing export: fail loudly
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def call(name, *args):
    """Invoke ``name`` and raise :class:`GarageAmdError` on a negative status."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.ga_last_error().decode('utf-8', 'replace')
        raise GarageAmdError('{} failed ({}): {}'.format(name, rc, msg))


def dptr(t):
    """Device (or host) address of a torch tensor / None -> NULL."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    """Current HIP stream of torch as a ``void*``."""
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
