"""VPG and PPO with garage's ``RLAlgorithm`` surface, device resident.

Mirrors ``garage.torch.algos.VPG`` / ``PPO`` (``torch/algos/vpg.py:17-455``,
``torch/algos/ppo.py:8-132``): same constructor keywords and defaults, the same
private method names (``_train_once``, ``_compute_advantage``, ``_train``,
``_train_policy``, ``_train_value_function`` ...), the same order of operations
inside an iteration and the same nine logged scalars.  One iteration never
leaves the GPU except for the episode lengths and the logged scalars.
"""
import collections
import math

import numpy as np
import torch

from garage_amd import logger
from garage_amd._dtypes import (DeviceEpisodeBatch, StepType, is_discrete,
                                step_types_as_uint8)
from garage_amd._lib import call, dptr, stream_ptr
from garage_amd.engine import (HALF_LOG_2PI, center_advantages, gae_scan,
                               pad_rows, reduction_workspace, round4)
from garage_amd.optimizers import OptimizerWrapper, data_parallel_plan
from garage_amd.policies import GaussianMLPPolicy


class _OldPolicy:
    """Parameter snapshot standing in for ``copy.deepcopy(policy)`` (vpg.py:107)."""

    def __init__(self, policy):
        self.params = policy.net.params.clone()

    def sync(self, policy):
        self.params.copy_(policy.net.params)


class VPG:
    """Vanilla Policy Gradient (``torch/algos/vpg.py:17-455``)."""

    def __init__(self,
                 env_spec,
                 policy,
                 value_function,
                 sampler,
                 policy_optimizer=None,
                 vf_optimizer=None,
                 num_train_per_epoch=1,
                 discount=0.99,
                 gae_lambda=1,
                 center_adv=True,
                 positive_adv=False,
                 policy_ent_coeff=0.0,
                 use_softplus_entropy=False,
                 stop_entropy_gradient=False,
                 entropy_method='no_entropy'):
        self._discount = discount
        self.policy = policy
        self.max_episode_length = env_spec.max_episode_length
        self._value_function = value_function
        self._gae_lambda = gae_lambda
        self._center_adv = center_adv
        self._positive_adv = positive_adv
        self._policy_ent_coeff = policy_ent_coeff
        self._use_softplus_entropy = use_softplus_entropy
        self._stop_entropy_gradient = stop_entropy_gradient
        self._entropy_method = entropy_method
        self._n_samples = num_train_per_epoch
        self._env_spec = env_spec
        self._maximum_entropy = (entropy_method == 'max')
        self._entropy_regularzied = (entropy_method == 'regularized')
        self._check_entropy_configuration(entropy_method, center_adv,
                                          stop_entropy_gradient,
                                          policy_ent_coeff)
        self._episode_reward_mean = collections.deque(maxlen=100)
        self._sampler = sampler
        if getattr(policy, 'kind', None) not in ('gaussian', 'categorical'):
            raise NotImplementedError(
                'garage_amd.algos needs a garage_amd GaussianMLPPolicy or '
                'CategoricalMLPPolicy')
        self._policy_optimizer = policy_optimizer or OptimizerWrapper(
            torch.optim.Adam, policy)
        self._vf_optimizer = vf_optimizer or OptimizerWrapper(
            torch.optim.Adam, value_function)
        self._old_policy = _OldPolicy(policy)
        self._algo_id = 1  # VPG objective in the loss kernel
        self._lr_clip_range = 0.0
        self._comm = None  # set by garage_amd.distributed.shard_algo
        self.last_tabular = {}

    @staticmethod
    def _check_entropy_configuration(entropy_method, center_adv,
                                     stop_entropy_gradient, policy_ent_coeff):
        """``vpg.py:109-125``."""
        if entropy_method not in ('max', 'regularized', 'no_entropy'):
            raise ValueError('Invalid entropy_method')
        if entropy_method == 'max':
            if center_adv:
                raise ValueError('center_adv should be False when '
                                 'entropy_method is max')
            if not stop_entropy_gradient:
                raise ValueError('stop_gradient should be True when '
                                 'entropy_method is max')
        if entropy_method == 'no_entropy':
            if policy_ent_coeff != 0.0:
                raise ValueError('policy_ent_coeff should be zero '
                                 'when there is no entropy method')

    @property
    def discount(self):
        return self._discount

    # -- helpers --------------------------------------------------------------
    def _to_device_batch(self, eps):
        """Accept any EpisodeBatch-like; upload host arrays when needed."""
        if isinstance(eps, DeviceEpisodeBatch):
            return eps
        dev = self.policy.device
        lengths = np.asarray(eps.lengths).astype(np.int64)
        off = np.concatenate([[0], np.cumsum(lengths)])
        S = int(off[-1])
        obs = np.asarray(eps.observations, dtype=np.float32).reshape(S, -1)
        acts = np.asarray(eps.actions, dtype=np.float32).reshape(S, -1)
        last = np.asarray(eps.last_observations,
                          dtype=np.float32).reshape(len(lengths), -1)
        st = step_types_as_uint8(eps.step_types)
        return DeviceEpisodeBatch(
            eps.env_spec, lengths=lengths, obs_dev=pad_rows(obs),
            last_obs_dev=pad_rows(last), actions_dev=pad_rows(acts),
            rewards_dev=torch.from_numpy(
                np.asarray(eps.rewards, dtype=np.float32)).to(dev),
            step_types_dev=torch.from_numpy(st).to(dev),
            ep_off_dev=torch.from_numpy(off).to(dev),
            discrete=is_discrete(eps.env_spec.action_space))

    def _entropy_value(self):
        """Entropy of the Gaussian policy: state independent (scalar log-std).

        ``Independent(Normal).entropy() = A * (0.5 + 0.5 log 2pi + log_std)``,
        optionally through softplus (``vpg.py:408-432``).
        """
        A = self.policy.net.out_dim
        s = np.float32(self.policy.clamped_log_std())
        ent = np.float32(A) * (np.float32(0.5) + np.float32(HALF_LOG_2PI) + s)
        if self._use_softplus_entropy:
            ent = np.float32(ent if ent > 20 else math.log1p(math.exp(ent)))
        return float(ent)

    def _ent_flags(self):
        return (int(self._entropy_regularzied) |
                (int(self._use_softplus_entropy) << 1) |
                (int(self._stop_entropy_gradient) << 2))

    def _policy_loss_pass(self, batch, adv, old_ll, M, idx, params=None,
                          want_grad=False, ll_out=None, obs=None,
                          ent_out=None, ent_sum=None, head=None):
        """Forward + fused loss (+ gradient seed) over ``M`` rows.

        Returns ``(loss, head, dout)``; ``head`` is the policy MLP output (means
        for the Gaussian policy, class scores for the categorical one).
        """
        pol = self.policy
        net = pol.net
        saved = None
        if params is not None:  # evaluate with the old policy's parameters
            saved, net.params = net.params, params
        obs = batch.obs_dev if obs is None else obs
        try:
            # training passes compute the head layer inside the loss kernel,
            # exactly as the native epoch loop does (same kernels, same bits)
            fused = (want_grad and head is None and pol.kind == 'gaussian'
                     and ent_out is None and getattr(self, 'fuse_head', False)
                     and net.head_fusable())
            if head is None and not fused:
                # (outputs only unless a backward pass follows)
                head = net.forward(obs, M, row_idx=idx, keep_acts=want_grad)
            dout = net.dout_view(M) if want_grad else None
            loss = torch.empty(1, dtype=torch.float32, device=net.device)
            algo = self._algo_id if old_ll is not None else 1
            slabs = dptr(net._slabs) if want_grad else None
            splits = int(net._splits) if want_grad else 0
            ws = dptr(reduction_workspace(net.device))
            if fused:
                H, Wh, bh = net.forward_hidden(obs, M, row_idx=idx)
                head = net.out_view(M)
                has_min, mn, has_max, mx = pol._std_args()
                call('ga_head_ppo_gaussian_loss_f32', dptr(H), H.stride(0),
                     dptr(Wh), H.stride(0), dptr(bh), int(net.dims[-2]),
                     dptr(head), head.stride(0), dptr(batch.actions_dev),
                     batch.actions_dev.stride(0), dptr(old_ll), dptr(adv),
                     dptr(idx), dptr(net.params[0:1]), has_min, mn, has_max,
                     mx, M, net.out_dim, algo, float(self._lr_clip_range),
                     float(self._policy_ent_coeff), self._ent_flags(),
                     dptr(dout), dout.stride(0), dptr(ll_out), dptr(loss),
                     slabs, net.n_flat, splits, ws, stream_ptr())
            elif pol.kind == 'gaussian':
                has_min, mn, has_max, mx = pol._std_args()
                call('ga_ppo_gaussian_loss_f32', dptr(head), head.stride(0),
                     dptr(batch.actions_dev), batch.actions_dev.stride(0),
                     dptr(old_ll), dptr(adv), dptr(idx), dptr(net.params[0:1]),
                     has_min, mn, has_max, mx, M, net.out_dim, algo,
                     float(self._lr_clip_range),
                     float(self._policy_ent_coeff), self._ent_flags(),
                     dptr(dout), dptr(ll_out), dptr(loss), slabs, net.n_flat,
                     splits, ws, stream_ptr())
            else:
                call('ga_ppo_categorical_loss_f32', dptr(head), head.stride(0),
                     dptr(batch.actions_dev), batch.actions_dev.stride(0),
                     dptr(old_ll), dptr(adv), dptr(idx), M, net.out_dim,
                     int(pol.double_softmax), algo, float(self._lr_clip_range),
                     float(self._policy_ent_coeff), self._ent_flags(),
                     dptr(dout), dptr(ll_out), dptr(ent_out), dptr(loss),
                     dptr(ent_sum), slabs, net.n_flat, splits, ws,
                     stream_ptr())
        finally:
            if saved is not None:
                net.params = saved
        return loss, head, dout

    def _value_loss_pass(self, batch, returns, M, idx, want_grad=False,
                         v=None):
        vf = self._value_function
        net = vf.net
        dout = net.dout_view(M) if want_grad else None
        loss = torch.empty(1, dtype=torch.float32, device=net.device)
        if (want_grad and v is None and getattr(self, 'fuse_head', False)
                and net.head_fusable()):
            H, Wh, bh = net.forward_hidden(batch.obs_dev, M, row_idx=idx)
            v = net.out_view(M)
            call('ga_head_gaussian_nll_loss_f32', dptr(H), H.stride(0),
                 dptr(Wh), dptr(bh), int(net.dims[-2]), dptr(v), v.stride(0),
                 dptr(returns), dptr(idx), dptr(net.params[0:1]), M,
                 dptr(dout), dout.stride(0), dptr(loss), dptr(net._slabs),
                 net.n_flat, int(net._splits),
                 dptr(reduction_workspace(net.device)), stream_ptr())
            return loss, v, dout
        if v is None:
            v = net.forward(batch.obs_dev, M, row_idx=idx,
                            keep_acts=want_grad)
        call('ga_gaussian_nll_loss_f32', dptr(v), v.stride(0), dptr(returns),
             dptr(idx), dptr(net.params[0:1]), M, dptr(dout), dptr(loss),
             dptr(net._slabs) if want_grad else None, net.n_flat,
             int(net._splits) if want_grad else 0,
             dptr(reduction_workspace(net.device)), stream_ptr())
        return loss, v, dout

    def _kl_sum(self, head_old, s_old, head_new, s_new, M):
        out = torch.zeros(1, dtype=torch.float64, device=head_old.device)
        ws = dptr(reduction_workspace(out.device))
        if self.policy.kind == 'gaussian':
            call('ga_gaussian_kl_f32', dptr(head_old), dptr(head_new),
                 head_old.stride(0), M, self.policy.net.out_dim, float(s_old),
                 float(s_new), dptr(out), ws, stream_ptr())
        else:
            call('ga_categorical_kl_f32', dptr(head_old), dptr(head_new),
                 head_old.stride(0), M, self.policy.net.out_dim,
                 int(self.policy.double_softmax), dptr(out), ws, stream_ptr())
        return out

    def _categorical_entropy(self, batch, adv, S, zero_obs, n_pad, n_cells,
                             per_step=None):
        """Mean entropy over the padded (N, P) grid for a categorical policy
        (``vpg.py:408-432`` evaluated on padded observations, Q9); optionally
        the per-step entropies of the valid rows and the padding's entropy."""
        dev = adv.device
        tot = torch.zeros(1, dtype=torch.float64, device=dev)
        pad = torch.zeros(1, dtype=torch.float64, device=dev)
        self._policy_loss_pass(batch, adv, None, S, None, ent_out=per_step,
                               ent_sum=tot)
        self._policy_loss_pass(batch, adv, None, 1, None, obs=zero_obs,
                               ent_sum=pad)
        tot, pad = float(tot.item()), float(pad.item())
        return (tot + n_pad * pad) / n_cells, pad

    def _allreduce(self, tensor, op='sum'):
        if self._comm is not None:
            self._comm.all_reduce(tensor, op)

    def _padded_cells(self, lengths):
        """N * P of the reference's padded tensors (pad_batch_array widening)."""
        P = self.max_episode_length
        longest = int(lengths.max())
        P = longest if P is None else max(int(P), longest)
        return len(lengths) * P, P

    # -- one iteration (vpg.py:136-206) -----------------------------------------
    def _train_once(self, itr, eps):
        batch = self._to_device_batch(eps)
        pol, vf = self.policy, self._value_function
        dev = pol.device
        S = batch.n_samples
        lengths = batch.lengths
        n_cells, P = self._padded_cells(lengths)
        n_pad = n_cells - S
        pol.net._workspace(S)
        vf.net._workspace(S)
        zero_obs = torch.zeros(1, batch.obs_dev.shape[1], device=dev)
        share = self._dp_setup(S)

        # baselines on every valid step and on the all-zero observation the
        # reference feeds through the padding (vpg.py:147,155-156; Q2)
        values = torch.empty(S, 1, dtype=torch.float32, device=dev)
        vf.net.forward(batch.obs_dev, S, out=values, keep_acts=False)
        v0 = float(vf.net.forward(zero_obs, 1)[0, 0].item())

        bonus, bonus_steps = 0.0, None
        gaussian = pol.kind == 'gaussian'
        if self._maximum_entropy:  # vpg.py:158-160, padded cells included
            if gaussian:
                bonus = self._policy_ent_coeff * self._entropy_value()
            else:
                # per-step entropies; the padding gets the zero-observation one
                bonus_steps = torch.empty(S, dtype=torch.float32, device=dev)
                dummy = torch.zeros(S, dtype=torch.float32, device=dev)
                _, h_pad = self._categorical_entropy(
                    batch, dummy, S, zero_obs, n_pad, n_cells, bonus_steps)
                bonus = self._policy_ent_coeff * h_pad
                bonus_steps.mul_(self._policy_ent_coeff).sub_(bonus)
        longest = int(lengths.max())
        if int(lengths.min()) == longest:
            # equal-length episodes: the packed batch IS an (N, L) matrix, no
            # offsets to chase (one dependent load less per wave)
            adv, returns = gae_scan(
                batch.rewards_dev.view(-1, longest), values.view(-1, longest),
                discount=self._discount, gae_lambda=self._gae_lambda,
                max_episode_length=P, v0=v0, bonus_const=bonus,
                bonus=None if bonus_steps is None else bonus_steps.view(
                    -1, longest))
            adv, returns = adv.view(-1), returns.view(-1)
        else:
            adv, returns = gae_scan(
                batch.rewards_dev, values.view(-1), discount=self._discount,
                gae_lambda=self._gae_lambda, max_episode_length=P,
                offsets=batch.ep_off_dev, max_len=longest, v0=v0,
                bonus_const=bonus, bonus=bonus_steps)
        self._normalise_advantages(adv)

        # ---- diagnostics before the update (vpg.py:168-173) -----------------
        s_old = self._clamped(self._old_policy.params) if gaussian else 0.0
        old_ll = torch.empty(S, dtype=torch.float32, device=dev)
        _, mean_old, _ = self._policy_loss_pass(
            batch, adv, None, S, None, params=self._old_policy.params,
            ll_out=old_ll)
        mean_old = mean_old.clone()
        saved = pol.net.params
        pol.net.params = self._old_policy.params
        mean_old_pad = pol.net.forward(zero_obs, 1).clone()
        pol.net.params = saved
        # The old policy equals the current one unless somebody changed the
        # parameters between iterations (vpg.py:201 syncs them): then the
        # "new" forward of LossBefore / KLBefore would recompute mean_old bit
        # for bit, so it is reused.  Likewise the baselines ARE the value
        # function's outputs for LossBefore.
        same = bool(torch.equal(self._old_policy.params, pol.net.params))
        loss_before, mean_new, _ = self._policy_loss_pass(
            batch, adv, old_ll, S, None, head=mean_old if same else None)
        kl_before = self._mean_kl(mean_old, mean_old_pad, s_old, mean_new,
                                  zero_obs, S, n_pad, n_cells)
        vf_before, _, _ = self._value_loss_pass(batch, returns, S, None,
                                                v=values)

        steps_before = (pol.net.adam_steps,
                        self._value_function.net.adam_steps)
        self._train(batch, adv, returns, old_ll)

        # ---- diagnostics after the update (vpg.py:178-184) -------------------
        loss_after, mean_new, _ = self._policy_loss_pass(
            batch, adv, old_ll, S, None)
        kl_after = self._mean_kl(mean_old, mean_old_pad, s_old, mean_new,
                                 zero_obs, S, n_pad, n_cells)
        vf_after, _, _ = self._value_loss_pass(batch, returns, S, None)
        if gaussian:
            entropy = self._entropy_value()
        else:
            entropy, _ = self._categorical_entropy(batch, adv, S, zero_obs,
                                                   n_pad, n_cells)

        scalars = torch.stack([loss_before[0], loss_after[0], vf_before[0],
                               vf_after[0]]).to(torch.float64)
        scalars = torch.cat([scalars, kl_before, kl_after])
        if self._comm is not None:
            # losses are per-sample means: weight by this rank's sample share;
            # the KLs are already global (see _mean_kl)
            scalars[:4] *= share
            scalars[4:] /= self._comm.world_size
            self._allreduce(scalars, 'sum')
        pl_b, pl_a, vl_b, vl_a, kl_b, kl_a = scalars.cpu().tolist()
        # the one-launch small-minibatch step raises the last workspace slot if
        # one of its grid barriers ever gave up (its results are then garbage)
        # (such a launch, and every later one, returns before any parameter is
        # written: the networks hold the last complete optimizer step)
        faulted = []
        for tag in (0, 1):
            ws = reduction_workspace(dev, tag)
            if float(ws[-1]) != 0.0:
                ws[-2:].zero_()  # re-arm the barrier words for the next call
                faulted.append(tag)
        if faulted:
            # Adam's step counts were advanced on the host for steps the device
            # skipped: put them back to the iteration's start, so that a caller
            # who catches this and carries on (ga_set_small_step(0)) does not
            # train with bias corrections of steps that never happened
            pol.net.adam_steps, self._value_function.net.adam_steps = \
                steps_before
            raise RuntimeError(
                'ga_small_step: a grid barrier timed out (reduction workspace '
                'tag(s) {}: 0 = the main stream\'s network, 1 = the side '
                'stream\'s); the optimizer steps from that point on were '
                'skipped (parameters and Adam moments are those of the last '
                'complete step, the step counts those of the start of this '
                'iteration). Disable the one-launch step with '
                'ga_set_small_step(0).'.format(faulted))
        tab = logger.tabular
        with tab.prefix(self.policy.name):
            tab.record('/LossBefore', pl_b)
            tab.record('/LossAfter', pl_a)
            tab.record('/dLoss', pl_b - pl_a)
            tab.record('/KLBefore', kl_b)
            tab.record('/KL', kl_a)
            tab.record('/Entropy', entropy)
        with tab.prefix(self._value_function.name):
            tab.record('/LossBefore', vl_b)
            tab.record('/LossAfter', vl_a)
            tab.record('/dLoss', vl_b - vl_a)
        self.last_tabular = {
            'policy/LossBefore': pl_b, 'policy/LossAfter': pl_a,
            'policy/dLoss': pl_b - pl_a, 'policy/KLBefore': kl_b,
            'policy/KL': kl_a, 'policy/Entropy': entropy,
            'vf/LossBefore': vl_b, 'vf/LossAfter': vl_a,
            'vf/dLoss': vl_b - vl_a,
        }
        self.last_tensors = {'advantages': adv, 'returns': returns,
                             'values': values.view(-1), 'v0': v0}

        self._old_policy.sync(self.policy)  # vpg.py:201
        undiscounted = self._log_performance(itr, batch, returns)
        return np.mean(undiscounted)

    def _clamped(self, params):
        """Log std of the policy for the parameters in ``params``."""
        return self.policy.log_std_of(float(params[0].item()))[0]

    def _mean_kl(self, mean_old, mean_old_pad, s_old, mean_new, zero_obs, S,
                 n_pad, n_cells):
        """``vpg.py:381-406`` over the padded (N, P) grid (Q9): valid rows from
        the device sum, padded rows (all identical) in closed form."""
        s_new = (self.policy.clamped_log_std()
                 if self.policy.kind == 'gaussian' else 0.0)
        total = self._kl_sum(mean_old, s_old, mean_new, s_new, S)
        if n_pad > 0:
            new_pad = self.policy.net.forward(zero_obs, 1).clone()
            total = total + n_pad * self._kl_sum(mean_old_pad, s_old, new_pad,
                                                 s_new, 1)
        if self._comm is not None:
            cells = torch.tensor([float(n_cells)], dtype=torch.float64,
                                 device=total.device)
            self._allreduce(total, 'sum')
            self._allreduce(cells, 'sum')
            return total / cells  # identical on every rank
        return total / n_cells

    def _dp_setup(self, S):
        """Data parallel bookkeeping of one iteration: every rank learns all
        sample counts (one tiny collective), from which follow this rank's
        share of the global batch and a common number of minibatches per pass.
        Returns the share (1.0 in a single process)."""
        if self._comm is None:
            return 1.0
        counts = self._comm.all_gather_int(S, device=self.policy.device)
        share = S / float(sum(counts))
        for opt in (self._policy_optimizer, self._vf_optimizer):
            opt.dp_grad_scale = share
            mb = opt._minibatch_size
            opt.dp_minibatches = opt.dp_grad_scales = None
            if mb is None:
                continue
            opt.dp_minibatches, opt.dp_grad_scales = data_parallel_plan(
                counts, mb, self._comm.rank)
        return share

    def _normalise_advantages(self, adv):
        """``vpg.py:371-377`` (global moments when the batch is sharded)."""
        hook = None
        if self._comm is not None:
            hook = self._allreduce
        center_advantages(adv, center=self._center_adv,
                          positive=self._positive_adv, allreduce=hook)

    def _compute_advantage(self, rewards, valids, baselines):
        """``vpg.py:349-379`` on padded ``(N, P)`` device tensors -> packed."""
        rewards = torch.as_tensor(rewards).to(self.policy.device,
                                              torch.float32).contiguous()
        baselines = torch.as_tensor(baselines).to(self.policy.device,
                                                  torch.float32).contiguous()
        adv, _ = gae_scan(rewards, baselines, discount=self._discount,
                          gae_lambda=self._gae_lambda,
                          max_episode_length=rewards.shape[1])
        flat = torch.cat([adv[i, :int(v)] for i, v in enumerate(valids)])
        self._normalise_advantages(flat)
        return flat

    # -- the reference's evaluation helpers, same names and argument order ----
    # (vpg.py:295-347,381-455; ppo.py:96-132; trpo.py:93-119).  They take host or
    # device tensors, run the same kernels as the training path and return
    # device tensors WITHOUT autograd history: gradients exist only inside
    # _train_policy / _train_value_function, so callers that differentiate
    # through these (MAML's inner loop) are out of scope.
    class _Rows:
        """The two fields of a batch the loss pass reads."""

        def __init__(self, obs_dev, actions_dev):
            self.obs_dev, self.actions_dev = obs_dev, actions_dev

    def _rows(self, obs, actions=None):
        obs = torch.as_tensor(obs)
        lead = tuple(obs.shape[:-1]) if obs.dim() > 1 else (obs.shape[0], )
        O = self.policy.net.in_dim
        obs_dev = pad_rows(obs.reshape(-1, O))
        act_dev = None
        if actions is not None:
            actions = torch.as_tensor(actions)
            act_dev = pad_rows(actions.reshape(obs_dev.shape[0], -1))
        return self._Rows(obs_dev, act_dev), obs_dev.shape[0], lead

    def _log_likelihoods(self, rows, M, params=None):
        ll = torch.empty(M, dtype=torch.float32, device=self.policy.device)
        zeros = torch.zeros(M, dtype=torch.float32, device=self.policy.device)
        self._policy_loss_pass(rows, zeros, None, M, None, params=params,
                               ll_out=ll)
        return ll

    def _compute_objective(self, advantages, obs, actions, rewards):
        """``vpg.py:434-455`` (``ppo.py:96-132`` / ``trpo.py:93-119`` by
        ``_algo_id``): per-sample objective values, shape ``(M,)``."""
        del rewards
        rows, M, _ = self._rows(obs, actions)
        adv = torch.as_tensor(advantages).to(self.policy.device,
                                             torch.float32).reshape(-1)
        new_ll = self._log_likelihoods(rows, M)
        if self._algo_id == 1:
            return new_ll * adv
        old_ll = self._log_likelihoods(rows, M, params=self._old_policy.params)
        ratio = (new_ll - old_ll).exp()
        if self._algo_id == 2:
            return ratio * adv
        clipped = torch.clamp(ratio, min=1 - self._lr_clip_range,
                              max=1 + self._lr_clip_range)
        return torch.min(ratio * adv, clipped * adv)

    def _compute_policy_entropy(self, obs):
        """``vpg.py:408-432``: entropies with the leading shape of ``obs``."""
        rows, M, lead = self._rows(obs)
        dev = self.policy.device
        if self.policy.kind == 'gaussian':
            return torch.full(lead, self._entropy_value(), dtype=torch.float32,
                              device=dev)
        ent = torch.empty(M, dtype=torch.float32, device=dev)
        rows.actions_dev = torch.zeros(M, 4, dtype=torch.float32, device=dev)
        self._policy_loss_pass(rows, torch.zeros(M, device=dev), None, M, None,
                               ent_out=ent)
        return ent.reshape(lead)

    def _compute_loss_with_adv(self, obs, actions, rewards, advantages):
        """``vpg.py:324-347``: the scalar the policy step minimises."""
        del rewards
        rows, M, _ = self._rows(obs, actions)
        adv = torch.as_tensor(advantages).to(self.policy.device,
                                             torch.float32).reshape(-1)
        old_ll = None
        if self._algo_id != 1:
            old_ll = self._log_likelihoods(rows, M,
                                           params=self._old_policy.params)
        loss, _, _ = self._policy_loss_pass(rows, adv.contiguous(), old_ll, M,
                                            None)
        return loss[0]

    def _compute_loss(self, obs, actions, rewards, valids, baselines):
        """``vpg.py:295-322`` on padded ``(N, P, ...)`` inputs."""
        from garage_amd.functions import filter_valids
        obs, actions = torch.as_tensor(obs), torch.as_tensor(actions)
        rewards = torch.as_tensor(rewards)
        obs_flat = torch.cat(filter_valids(obs, valids))
        actions_flat = torch.cat(filter_valids(actions, valids))
        rewards_flat = torch.cat(filter_valids(rewards, valids))
        adv = self._compute_advantage(rewards, valids, baselines)
        return self._compute_loss_with_adv(obs_flat, actions_flat,
                                           rewards_flat, adv)

    def _compute_kl_constraint(self, obs):
        """``vpg.py:381-406``: mean KL(old || new) over every row of ``obs``."""
        rows, M, _ = self._rows(obs)
        pol = self.policy
        net = pol.net
        saved, net.params = net.params, self._old_policy.params
        try:
            head_old = net.forward(rows.obs_dev, M).clone()
        finally:
            net.params = saved
        head_new = net.forward(rows.obs_dev, M)
        s_old = s_new = 0.0
        if pol.kind == 'gaussian':
            s_old = self._clamped(self._old_policy.params)
            s_new = pol.clamped_log_std()
        kl = self._kl_sum(head_old, s_old, head_new, s_new, M)
        return (kl[0] / M).to(torch.float32)

    # -- the update (vpg.py:230-293) -------------------------------------------
    def _train(self, batch, adv, returns, old_ll):
        S = batch.n_samples
        if self._native_update_ok():
            if getattr(self, 'overlap_updates', True):
                self._train_native_pair(batch, adv, returns, old_ll)
            else:
                self._train_native_serial(batch, adv, returns, old_ll)
            return
        for idx in self._policy_optimizer.minibatch_indices(S):
            self._train_policy(batch, adv, old_ll, idx)
        for idx in self._vf_optimizer.minibatch_indices(S):
            self._train_value_function(batch, returns, idx)

    def _native_update_ok(self):
        """The C++ epoch loop (``ga_update_epoch``) replaces the Python loop
        unless a subclass hooks the per-minibatch methods or the ranks exchange
        gradients through a backend the library cannot call (gloo)."""
        cls = type(self)
        if (cls._train_policy is not VPG._train_policy
                or cls._train_value_function is not VPG._train_value_function):
            return False
        for opt in (self._policy_optimizer, self._vf_optimizer):
            if opt.grad_hook is not None and getattr(opt, 'native_comm',
                                                     None) is None:
                return False
            # the native loops fuse torch's default-shaped Adam; any other
            # torch.optim class steps from the per-minibatch loop
            if not getattr(opt, 'default_adam', True):
                return False
        return True

    def _update_args(self, opt, module, kind, batch, adv, returns, old_ll,
                     tag, rows=None):
        """``ga_update_args`` of one network (kept alive by the caller);
        ``rows``: size the workspaces for minibatches of that many rows."""
        from garage_amd import _lib
        import ctypes as C
        # head layer inside the loss kernel (opt-in; same switch as the Python
        # minibatch loop so both produce the same bits)
        _lib.load().ga_set_fused_head_loss(
            int(bool(getattr(self, 'fuse_head', False))))
        net = module.net
        S = batch.n_samples
        mb = opt.local_minibatch_size(S)
        if rows is None:
            rows = S if mb is None else min(S, mb)
        net._workspace(rows)
        dev = net.device
        a = _lib.UpdateArgs()
        a.desc = C.pointer(net._desc)
        a.params, a.grads = net.params.data_ptr(), net.grads.data_ptr()
        a.exp_avg = net.exp_avg.data_ptr()
        a.exp_avg_sq = net.exp_avg_sq.data_ptr()
        a.n_flat = net.n_flat
        a.acts, a.dacts = net._acts.data_ptr(), net._dacts.data_ptr()
        a.out, a.dout = net._out.data_ptr(), net._dout.data_ptr()
        a.ldo = net.ld_out
        a.slabs, a.max_splits = net._slabs.data_ptr(), int(net._splits)
        h = opt._hyper
        # (Adam's constants: unused by the gradients-only phase that every other
        # torch.optim class takes)
        betas = h.get('betas', (0.9, 0.999))
        a.lr, a.beta1, a.beta2, a.eps = (float(h['lr']), float(betas[0]),
                                         float(betas[1]),
                                         float(h.get('eps', 1e-8)))
        a.learn_std = int(getattr(module, '_learn_std', True))
        a.X, a.ldx, a.S = (batch.obs_dev.data_ptr(), batch.obs_dev.stride(0),
                           S)
        a.mb = 0 if mb is None else int(mb)
        a.kind = kind
        if kind == 0:
            a.actions = batch.actions_dev.data_ptr()
            a.lda = batch.actions_dev.stride(0)
            a.old_ll, a.adv = old_ll.data_ptr(), adv.data_ptr()
            if module.kind == 'gaussian':
                a.has_min, a.min_log_std, a.has_max, a.max_log_std = \
                    module._std_args()
            else:
                a.kind = 2
                a.double_softmax = int(module.double_softmax)
            a.algo = self._algo_id
            a.clip = float(self._lr_clip_range)
            a.ent_coeff = float(self._policy_ent_coeff)
            a.ent_flags = self._ent_flags()
        else:
            a.returns = returns.data_ptr()
        scratch = torch.empty(1, dtype=torch.float32, device=dev)
        a.loss_scratch = scratch.data_ptr()
        a.workspace = reduction_workspace(dev, tag).data_ptr()
        comm = getattr(opt, 'native_comm', None)
        keep = [scratch]
        # fused step kernels (last hidden layer + head + loss in one launch, ...)
        # for minibatches a few tiles and up; VPG's single full-batch step of a
        # million rows would need a scratch of its own size for nothing
        part = net.train_partials(rows) if rows <= (1 << 17) else None
        if part is not None:
            a.partials, a.partials_floats = part.data_ptr(), part.numel()
        if comm is not None:
            a.comm, a.world = comm.handle, comm.world_size
            a.grad_scale = float(opt.dp_grad_scale)
        n_mb = len(opt.minibatch_bounds(S)) - 1
        if mb is not None and opt.dp_minibatches:
            a.n_mb = n_mb  # even split: the same count on every rank
            if opt.dp_grad_scales is not None:
                scales = np.ascontiguousarray(opt.dp_grad_scales, np.float32)
                assert scales.size == n_mb
                a.grad_scales_host = scales.ctypes.data
                keep.append(scales)
        return a, keep, n_mb

    def _train_native_serial(self, batch, adv, returns, old_ll):
        """Policy pass then value pass on the current stream (the reference's
        order, ``vpg.py:244-248``); ``overlap_updates = False`` selects it."""
        import ctypes as C
        S = batch.n_samples
        for opt, module, kind in ((self._policy_optimizer, self.policy, 0),
                                  (self._vf_optimizer, self._value_function,
                                   1)):
            a, keep, n_mb = self._update_args(opt, module, kind, batch, adv,
                                              returns, old_ll, 0)
            for perm in opt.epoch_permutations(S):
                a.perm = None if perm is None else perm.data_ptr()
                a.step0 = module.net.adam_steps
                call('ga_update_epoch', C.byref(a), stream_ptr())
                module.net.adam_steps += n_mb
            del keep

    def _train_native_pair(self, batch, adv, returns, old_ll):
        """Policy and value-function passes, interleaved on two HIP streams.

        The reference finishes the policy before it starts the value function
        (``vpg.py:244-248``); the two share no written state, so
        overlapping them changes nothing but the wall time.  The host-side
        permutation draws keep the reference's order: every policy shuffle
        happens before the first value-function shuffle.
        """
        import ctypes as C
        S = batch.n_samples
        popt, vopt = self._policy_optimizer, self._vf_optimizer
        pa, keep_p, n_p = self._update_args(popt, self.policy, 0, batch, adv,
                                            returns, old_ll, 0)
        va, keep_v, n_v = self._update_args(vopt, self._value_function, 1,
                                            batch, adv, returns, old_ll, 1)
        p_perms = list(popt.epoch_permutations(S))
        v_perms = list(vopt.epoch_permutations(S))
        main = torch.cuda.current_stream()
        if getattr(self, '_side_stream', None) is None:
            self._side_stream = torch.cuda.Stream()
        side = self._side_stream
        side.wait_stream(main)  # inputs (adv, returns, perms) are ready
        pnet, vnet = self.policy.net, self._value_function.net
        s_main = C.c_void_p(main.cuda_stream)
        s_side = C.c_void_p(side.cuda_stream)
        for e in range(max(len(p_perms), len(v_perms))):
            has_p, has_v = e < len(p_perms), e < len(v_perms)
            if has_p:
                pp = p_perms[e]
                pa.perm = None if pp is None else pp.data_ptr()
                pa.step0 = pnet.adam_steps
            if has_v:
                vp = v_perms[e]
                va.perm = None if vp is None else vp.data_ptr()
                va.step0 = vnet.adam_steps
            if has_p and has_v:
                call('ga_update_epoch_pair', C.byref(pa), s_main, C.byref(va),
                     s_side)
            elif has_p:
                call('ga_update_epoch', C.byref(pa), s_main)
            else:
                call('ga_update_epoch', C.byref(va), s_side)
            if has_p:
                pnet.adam_steps += n_p
            if has_v:
                vnet.adam_steps += n_v
        main.wait_stream(side)
        del keep_p, keep_v

    def _train_policy(self, batch, adv, old_ll, idx):
        """``vpg.py:250-272``: one optimizer step on the rows ``idx``."""
        return self._minibatch_step(self._policy_optimizer, self.policy, 0,
                                    batch, adv, None, old_ll, idx)

    def _train_value_function(self, batch, returns, idx):
        """``vpg.py:274-293``."""
        return self._minibatch_step(self._vf_optimizer, self._value_function,
                                    1, batch, None, returns, None, idx)

    def _minibatch_step(self, opt, module, kind, batch, adv, returns, old_ll,
                        idx):
        """One step through the same entry point as the native epoch loop
        (``ga_update_epoch`` with a pass of one minibatch), so both produce the
        same bits.  With a gradient hook (data parallel over a backend the
        library cannot call) the native part stops at the reduced gradient
        (``phase = 1``); the exchange and Adam follow here."""
        import ctypes as C
        net = module.net
        M = batch.n_samples if idx is None else int(idx.numel())
        a, keep, _ = self._update_args(opt, module, kind, batch, adv, returns,
                                       old_ll, 0, rows=M)
        a.S, a.mb, a.n_mb = M, M, 0
        a.perm = None if idx is None else idx.data_ptr()
        if idx is None:
            a.S = batch.n_samples
        a.grad_scales_host = None
        a.comm = None
        hook = opt.grad_hook
        generic = not getattr(opt, 'default_adam', True)
        a.phase = 1 if (hook is not None or generic) else 0
        if hook is not None:
            a.grad_scale = float(opt.dp_grad_scale
                                 if opt._cur_grad_scale is None
                                 else opt._cur_grad_scale)
        elif generic:
            a.grad_scale = 1.0
        loss = torch.empty(1, dtype=torch.float32, device=net.device)
        a.losses = loss.data_ptr()
        a.step0 = net.adam_steps
        call('ga_update_epoch', C.byref(a), stream_ptr())
        if hook is not None:
            hook(net.grads)
        if hook is not None or generic:
            opt.apply_step()
        else:
            net.adam_steps += 1
        del keep
        return loss

    # -- log_performance (_functions.py:233-275) --------------------------------
    def _log_performance(self, itr, batch, returns):
        dev = returns.device
        off = batch.ep_off_dev
        N = len(batch.lengths)
        sums = torch.empty(N, dtype=torch.float64, device=dev)
        call('ga_episode_sums_f32', dptr(batch.rewards_dev), dptr(off), N,
             dptr(sums), stream_ptr())
        first = returns[off[:-1]].to(torch.float64)
        last_st = batch.step_types_dev[off[1:] - 1]
        host = torch.cat([sums, first,
                          (last_st == int(StepType.TERMINAL)).to(
                              torch.float64)]).cpu().numpy()
        undiscounted, discounted, term = host[:N], host[N:2 * N], host[2 * N:]
        tab = logger.tabular
        with tab.prefix('Evaluation/'):
            tab.record('Iteration', itr)
            tab.record('NumEpisodes', N)
            tab.record('AverageDiscountedReturn', np.mean(discounted))
            tab.record('AverageReturn', np.mean(undiscounted))
            tab.record('StdReturn', np.std(undiscounted))
            tab.record('MaxReturn', np.max(undiscounted))
            tab.record('MinReturn', np.min(undiscounted))
            tab.record('TerminationRate', np.mean(term))
            if 'success' in batch.env_infos:  # CPU envs that report it
                lengths = np.asarray(batch.lengths, dtype=np.int64)
                flags = np.asarray(batch.env_infos['success']).reshape(
                    int(lengths.sum()), -1).any(axis=1)
                starts = np.concatenate([[0], np.cumsum(lengths)[:-1]])
                tab.record('SuccessRate', np.mean(
                    np.logical_or.reduceat(flags, starts).astype(np.float64)))
        self.last_performance = {
            'NumEpisodes': N,
            'AverageDiscountedReturn': float(np.mean(discounted)),
            'AverageReturn': float(np.mean(undiscounted)),
            'StdReturn': float(np.std(undiscounted)),
            'MaxReturn': float(np.max(undiscounted)),
            'MinReturn': float(np.min(undiscounted)),
            'TerminationRate': float(np.mean(term)),
        }
        return list(undiscounted)

    def train(self, trainer):
        """``vpg.py:208-228``."""
        last_return = None
        for _ in trainer.step_epochs():
            for _ in range(self._n_samples):
                eps = trainer.obtain_episodes(trainer.step_itr)
                last_return = self._train_once(trainer.step_itr, eps)
                trainer.step_itr += 1
        return last_return

    # -- snapshots: the sampler's workers are rebuilt on load ---------------------
    def __getstate__(self):
        state = self.__dict__.copy()
        state['_old_policy'] = self._old_policy.params.cpu().numpy()
        for k in ('last_tensors', 'last_cg', '_side_stream'):
            state.pop(k, None)  # device handles / scratch, rebuilt on demand
        state['_comm'] = None
        return state

    def __setstate__(self, state):
        old = state.pop('_old_policy')
        self.__dict__.update(state)
        self._old_policy = _OldPolicy(self.policy)
        self._old_policy.params.copy_(torch.from_numpy(old))


class PPO(VPG):
    """Proximal Policy Optimization (``torch/algos/ppo.py:8-132``)."""

    def __init__(self,
                 env_spec,
                 policy,
                 value_function,
                 sampler,
                 policy_optimizer=None,
                 vf_optimizer=None,
                 lr_clip_range=2e-1,
                 num_train_per_epoch=1,
                 discount=0.99,
                 gae_lambda=0.97,
                 center_adv=True,
                 positive_adv=False,
                 policy_ent_coeff=0.0,
                 use_softplus_entropy=False,
                 stop_entropy_gradient=False,
                 entropy_method='no_entropy'):
        if policy_optimizer is None:
            policy_optimizer = OptimizerWrapper(
                (torch.optim.Adam, dict(lr=2.5e-4)), policy,
                max_optimization_epochs=10, minibatch_size=64)
        if vf_optimizer is None:
            vf_optimizer = OptimizerWrapper(
                (torch.optim.Adam, dict(lr=2.5e-4)), value_function,
                max_optimization_epochs=10, minibatch_size=64)
        super().__init__(env_spec=env_spec, policy=policy,
                         value_function=value_function, sampler=sampler,
                         policy_optimizer=policy_optimizer,
                         vf_optimizer=vf_optimizer,
                         num_train_per_epoch=num_train_per_epoch,
                         discount=discount, gae_lambda=gae_lambda,
                         center_adv=center_adv, positive_adv=positive_adv,
                         policy_ent_coeff=policy_ent_coeff,
                         use_softplus_entropy=use_softplus_entropy,
                         stop_entropy_gradient=stop_entropy_gradient,
                         entropy_method=entropy_method)
        self._lr_clip_range = lr_clip_range
        self._algo_id = 0  # clipped surrogate in the loss kernel


class TRPO(VPG):
    """Trust Region Policy Optimization (``torch/algos/trpo.py:9-144``).

    The policy step is garage's ``ConjugateGradientOptimizer``
    (``torch/optimizers/conjugate_gradient_optimizer.py``) on the device:
    gradient of the unclipped surrogate, 10 conjugate-gradient iterations on the
    KL constraint's Hessian, the step size ``sqrt(2 delta / s^T A s)`` and the
    backtracking line search.  The reference takes Hessian-vector products by
    double backward; here ``A v = J^T M (J v) + reg v`` with ``J v`` a tangent
    forward pass (``ga_mlp_jvp_f32``), ``M`` the Gaussian metric
    (``ga_fisher_seed_gaussian_f32``) or, for ``CategoricalMLPPolicy``, the
    categorical one through the head's softmaxes
    (``ga_fisher_seed_categorical_f32``), and ``J^T`` the ordinary backward pass --
    the same matrix, because the step starts at the old policy's parameters,
    where the KL's gradient with respect to the distribution vanishes and its
    Hessian is the Fisher matrix.  Both heads are pinned by real TRPO iterations
    of the reference with every conjugate-gradient product recorded
    (``tests/golden/trpo_train_once.npz``, ``trpo_categorical.npz``).
    """

    def __init__(self,
                 env_spec,
                 policy,
                 value_function,
                 sampler,
                 policy_optimizer=None,
                 vf_optimizer=None,
                 num_train_per_epoch=1,
                 discount=0.99,
                 gae_lambda=0.98,
                 center_adv=True,
                 positive_adv=False,
                 policy_ent_coeff=0.0,
                 use_softplus_entropy=False,
                 stop_entropy_gradient=False,
                 entropy_method='no_entropy'):
        from garage_amd.optimizers import ConjugateGradientOptimizer
        if policy_optimizer is None:
            policy_optimizer = OptimizerWrapper(
                (ConjugateGradientOptimizer, dict(max_constraint_value=0.01)),
                policy)
        if vf_optimizer is None:
            vf_optimizer = OptimizerWrapper(
                (torch.optim.Adam, dict(lr=2.5e-4)), value_function,
                max_optimization_epochs=10, minibatch_size=64)
        super().__init__(env_spec=env_spec, policy=policy,
                         value_function=value_function, sampler=sampler,
                         policy_optimizer=policy_optimizer,
                         vf_optimizer=vf_optimizer,
                         num_train_per_epoch=num_train_per_epoch,
                         discount=discount, gae_lambda=gae_lambda,
                         center_adv=center_adv, positive_adv=positive_adv,
                         policy_ent_coeff=policy_ent_coeff,
                         use_softplus_entropy=use_softplus_entropy,
                         stop_entropy_gradient=stop_entropy_gradient,
                         entropy_method=entropy_method)
        if policy.kind not in ('gaussian', 'categorical'):
            raise NotImplementedError(
                'garage_amd.algos.TRPO implements the Gaussian and the '
                'categorical MLP policy')
        hyper = self._policy_optimizer._hyper
        if hyper.get('kind') != 'cg':
            raise NotImplementedError(
                'TRPO needs OptimizerWrapper((ConjugateGradientOptimizer, '
                '{...}), policy)')
        if (self._policy_optimizer._minibatch_size is not None
                or self._policy_optimizer._max_optimization_epochs != 1):
            # later steps would start away from the old policy, where the KL
            # Hessian is no longer the Fisher matrix this class multiplies by
            raise NotImplementedError(
                'TRPO takes one full-batch policy step per iteration '
                '(trpo.py:64-66 default wrapper)')
        self._algo_id = 2  # unclipped surrogate in the loss kernel
        self.last_cg = {}
        self._trpo_share = 1.0  # this rank's share of the global batch

    # -- update (vpg.py:230-248 with the constrained policy step) ---------------
    def _train(self, batch, adv, returns, old_ll):
        import ctypes as C
        S = batch.n_samples
        opt, vf = self._vf_optimizer, self._value_function
        native = (type(self)._train_value_function
                  is VPG._train_value_function
                  and getattr(opt, 'default_adam', True)
                  and (opt.grad_hook is None
                       or getattr(opt, 'native_comm', None) is not None))
        if not native:
            self._train_policy(batch, adv, old_ll, None)
            for idx in opt.minibatch_indices(S):
                self._train_value_function(batch, returns, idx)
            return
        # One GPU: the value function's epochs go to a second stream and run under
        # the policy step (conjugate gradient + line search: full-batch kernels
        # with a host round trip per iteration).  The two share no written state
        # (trpo.py / vpg.py:244-248 finish the policy first; the policy step draws
        # no random numbers, so the value function's shuffles are the same), only
        # the wall time changes.  With a process group the order stays sequential:
        # two chains of collectives on two communicators are not worth the risk.
        overlap = (getattr(self, 'overlap_updates', True)
                   and self._comm is None)
        if not overlap:
            self._train_policy(batch, adv, old_ll, None)
        a, keep, n_mb = self._update_args(opt, vf, 1, batch, adv, returns,
                                          old_ll, 1 if overlap else 0)
        stream = stream_ptr()
        main = torch.cuda.current_stream()
        # Every epoch's permutation is drawn on the main stream BEFORE the side
        # stream's wait and stays referenced until the join below: a permutation
        # drawn lazily inside the loop would be read by the side stream without
        # an edge from the kernel that writes it, and its block could be handed
        # to a later epoch's draw while the side stream still gathers through it
        # (the same rule as _train_native_pair).
        perms = list(opt.epoch_permutations(S))
        if overlap:
            if getattr(self, '_side_stream', None) is None:
                self._side_stream = torch.cuda.Stream()
            # returns, baselines and the permutations are ready
            self._side_stream.wait_stream(main)
            stream = C.c_void_p(self._side_stream.cuda_stream)
        for perm in perms:
            a.perm = None if perm is None else perm.data_ptr()
            a.step0 = vf.net.adam_steps
            call('ga_update_epoch', C.byref(a), stream)
            vf.net.adam_steps += n_mb
        if overlap:
            self._train_policy(batch, adv, old_ll, None)
            main.wait_stream(self._side_stream)
        del keep, perms

    def _dot(self, a, b):
        """Host float of ``a . b`` (fp64 accumulation on the device)."""
        out = torch.empty(1, dtype=torch.float64, device=a.device)
        call('ga_dot_f32', dptr(a), dptr(b), a.numel(), dptr(out),
             stream_ptr())
        return float(out.item())

    def _fisher_vector_product(self, batch, M, vec, out):
        """``out = A vec``: Hessian of ``mean KL(old || new)`` at new == old,
        plus ``hvp_reg_coeff * vec`` (conjugate_gradient_optimizer.py:18-66)."""
        pol = self.policy
        net = pol.net
        hyper = self._policy_optimizer._hyper
        tout = net.jvp(batch.obs_dev, M, vec)
        dout = net.dout_view(M)
        if pol.kind == 'gaussian':
            has_min, mn, has_max, mx = pol._std_args()
            call('ga_fisher_seed_gaussian_f32', dptr(tout), tout.stride(0), M,
                 net.out_dim, dptr(net.params[0:1]), has_min, mn, has_max, mx,
                 dptr(dout), dout.stride(0), stream_ptr())
        else:
            # class scores of the forward pass the step started from (nothing
            # runs a forward between it and the conjugate-gradient products)
            scores = net.out_view(M)
            call('ga_fisher_seed_categorical_f32', dptr(scores),
                 scores.stride(0), dptr(tout), tout.stride(0), M, net.out_dim,
                 int(pol.double_softmax), dptr(dout), dout.stride(0),
                 stream_ptr())
        net.backward(batch.obs_dev, M, dout)
        # the seed divides by this rank's M: rescale to the global batch, then
        # sum the ranks' J^T M J v (vec is replicated, so every rank ends up
        # with the same product)
        net.reduce_grads(scale=self._trpo_share)
        g = net.grads
        if self._comm is not None:
            self._comm.all_reduce(g, 'sum')
        # the log-std block: d2/ds2 of sum_a [s - s_old + exp(2(s_old - s))/2]
        # = 2 A at s == s_old, through the clamp's pass-through gradient
        # (and the std parameterisation: log std = f(p) has d2 KL / dp2 =
        # f'(p)^2 d2 KL / ds2 there, the first derivative of the KL being zero)
        if pol.kind == 'gaussian':
            chain = pol.log_std_of(float(net.params[0].item()))[1]
            if not getattr(pol, '_learn_std', True):
                chain = 0.0
            g[0:1].copy_(vec[0:1] * (2.0 * net.out_dim * chain * chain))
        else:
            g[0:1].zero_()  # the flat layout's std slot: not a parameter here
        out.copy_(g)
        call('ga_axpby_f32', float(hyper['hvp_reg_coeff']), dptr(vec), 1.0,
             dptr(out), out.numel(), stream_ptr())
        return out

    def _train_policy(self, batch, adv, old_ll, idx):
        """``trpo.py:121-144`` + ``ConjugateGradientOptimizer.step``
        (``conjugate_gradient_optimizer.py:146-186,236-277``)."""
        assert idx is None
        pol = self.policy
        net = pol.net
        hyper = self._policy_optimizer._hyper
        M = batch.n_samples
        dev = net.device
        n = net.n_flat
        # data parallel: this rank's share of the global batch (1.0 alone); the
        # global mean loss / gradient / KL are share-weighted sums over ranks
        share, M_glob = 1.0, M
        if self._comm is not None:
            counts = self._comm.all_gather_int(M, device=dev)
            M_glob = int(sum(counts))
            share = M / float(M_glob)
        self._trpo_share = share
        # gradient of the surrogate loss (flat_loss_grads)
        loss0, mean_old, dout = self._policy_loss_pass(batch, adv, old_ll, M,
                                                       None, want_grad=True)
        mean_old = mean_old.clone()
        s_old = pol.clamped_log_std()
        net.backward(batch.obs_dev, M, dout)
        net.reduce_grads(scale=share)
        if pol.kind != 'gaussian' or not getattr(pol, '_learn_std', True):
            net.grads[0:1].zero_()
        if self._comm is not None:
            self._comm.all_reduce(net.grads, 'sum')
        b = net.grads.clone()

        # conjugate gradient (Demmel p. 312), conjugate_gradient_optimizer.py:69-104
        x = torch.zeros(n, dtype=torch.float32, device=dev)
        r = b.clone()
        p = b.clone()
        z = torch.empty(n, dtype=torch.float32, device=dev)
        rdotr = self._dot(r, r)
        for _ in range(int(hyper['cg_iters'])):
            self._fisher_vector_product(batch, M, p, z)
            v = rdotr / self._dot(p, z)
            call('ga_axpby_f32', v, dptr(p), 1.0, dptr(x), n, stream_ptr())
            call('ga_axpby_f32', -v, dptr(z), 1.0, dptr(r), n, stream_ptr())
            newrdotr = self._dot(r, r)
            mu = newrdotr / rdotr
            call('ga_axpby_f32', 1.0, dptr(r), mu, dptr(p), n, stream_ptr())
            rdotr = newrdotr
            if rdotr < 1e-10:
                break
        x = torch.nan_to_num(x, nan=0.0, posinf=float('inf'),
                             neginf=float('-inf'))
        self._fisher_vector_product(batch, M, x, z)
        sAs = self._dot(x, z)
        with np.errstate(all='ignore'):
            step_size = float(np.sqrt(2.0 * hyper['max_constraint_value'] *
                                      (1. / (sAs + 1e-8))))
        if math.isnan(step_size):
            step_size = 1.
        descent = x * step_size
        self.last_cg = dict(grad=b, step_dir=x, descent_step=descent)

        # backtracking line search, conjugate_gradient_optimizer.py:236-277
        prev = net.params.clone()

        def global_mean(local_mean):
            if self._comm is None:
                return float(local_mean.item())
            t = local_mean.double().reshape(1) * share
            self._comm.all_reduce(t, 'sum')
            return float(t.item())

        loss_before = global_mean(loss0)
        accepted = -1
        loss = constraint = float('nan')
        for k in range(int(hyper['max_backtracks'])):
            ratio = float(hyper['backtrack_ratio'])**k
            net.params.copy_(prev)
            call('ga_axpby_f32', -ratio, dptr(descent), 1.0,
                 dptr(net.params), n, stream_ptr())
            l_new, mean_new, _ = self._policy_loss_pass(batch, adv, old_ll, M,
                                                        None)
            loss = global_mean(l_new)
            kl = self._kl_sum(mean_old, s_old, mean_new,
                              pol.clamped_log_std(), M)
            self._allreduce(kl)
            constraint = float(kl.item()) / float(M_glob)
            if (loss < loss_before
                    and constraint <= hyper['max_constraint_value']):
                accepted = k
                break
        if ((math.isnan(loss) or math.isnan(constraint)
             or loss >= loss_before
             or constraint >= hyper['max_constraint_value'])
                and not hyper['accept_violation']):
            logger.log('Line search condition violated. Rejecting the step!')
            if math.isnan(loss):
                logger.log('Violated because loss is NaN')
            if math.isnan(constraint):
                logger.log('Violated because constraint is NaN')
            if loss >= loss_before:
                logger.log('Violated because loss not improving')
            if constraint >= hyper['max_constraint_value']:
                logger.log('Violated because constraint is violated')
            net.params.copy_(prev)
            accepted = -1
        self.last_cg['accepted'] = accepted
        return loss0


__all__ = ['VPG', 'PPO', 'TRPO']
