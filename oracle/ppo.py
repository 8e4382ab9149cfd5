"""Oracle: one PPO / VPG training iteration on the CPU (torch autograd + Adam).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Restates
  * ``torch/algos/vpg.py:136-206``  VPG._train_once  (order of every step,
    which tensors are padded and which are packed, the nine logged scalars)
  * ``torch/algos/vpg.py:230-293``  _train / _train_policy / _train_value_function
    (ALL policy minibatches for all epochs first, then the value function)
  * ``torch/algos/vpg.py:324-347,381-432`` loss-with-entropy, KL, entropy
  * ``torch/algos/ppo.py:96-132``   clipped surrogate
  * ``torch/optimizers/optimizer_wrapper.py:22-63`` + ``_functions.py:25-65``
    (``torch.optim.Adam(module.parameters(), lr=...)``)
It is also the ``cpu_baseline`` (kind "port") timed by ``bench.py``.
"""
import copy
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from oracle import networks as nets
from oracle.batch import minibatch_index_stream, performance_stats
from oracle.returns import (filter_valids, padded_returns,
                            vpg_compute_advantage)


class OraclePPO:
    """CPU PPO (``algo='ppo'``) or VPG (``algo='vpg'``) with garage semantics."""

    def __init__(self,
                 policy_params,
                 value_params,
                 *,
                 max_episode_length,
                 algo='ppo',
                 policy_kind='gaussian',
                 double_softmax=True,
                 lr_clip_range=0.2,
                 discount=0.99,
                 gae_lambda=0.97,
                 center_adv=True,
                 positive_adv=False,
                 policy_ent_coeff=0.0,
                 use_softplus_entropy=False,
                 stop_entropy_gradient=False,
                 entropy_method='no_entropy',
                 policy_lr=2.5e-4,
                 vf_lr=2.5e-4,
                 max_optimization_epochs=10,
                 minibatch_size=64,
                 policy_optimizer=None,
                 vf_optimizer=None):
        self.policy = OrderedDict(
            (k, v.clone().detach()) for k, v in policy_params.items())
        self.value = OrderedDict(
            (k, v.clone().detach()) for k, v in value_params.items())
        for k in nets.trainable_keys(self.policy):
            self.policy[k].requires_grad_(True)
        for k in nets.trainable_keys(self.value):
            self.value[k].requires_grad_(True)
        self.old_policy = self._snapshot(self.policy)  # vpg.py:107
        self.P = max_episode_length
        self.algo = algo
        self.kind = policy_kind
        self.double_softmax = double_softmax
        self.clip = lr_clip_range
        self.discount = discount
        self.gae_lambda = gae_lambda
        self.center_adv = center_adv
        self.positive_adv = positive_adv
        self.ent_coeff = policy_ent_coeff
        self.softplus_entropy = use_softplus_entropy
        self.stop_entropy_gradient = stop_entropy_gradient
        # vpg.py:109-125 configuration checks.
        if entropy_method not in ('max', 'regularized', 'no_entropy'):
            raise ValueError('Invalid entropy_method')
        if entropy_method == 'max':
            if center_adv:
                raise ValueError('center_adv should be False when '
                                 'entropy_method is max')
            if not stop_entropy_gradient:
                raise ValueError('stop_gradient should be True when '
                                 'entropy_method is max')
        if entropy_method == 'no_entropy' and policy_ent_coeff != 0.0:
            raise ValueError('policy_ent_coeff should be zero '
                             'when there is no entropy method')
        self.max_entropy = entropy_method == 'max'
        self.reg_entropy = entropy_method == 'regularized'
        self.epochs = max_optimization_epochs
        self.mb = minibatch_size
        # make_optimizer (_functions.py:25-65): a torch.optim type or (type, kwargs);
        # the default is what ppo.py:65-76 builds
        def build(spec, params, lr):
            if spec is None:
                return torch.optim.Adam(params, lr=lr)
            if isinstance(spec, tuple):
                return spec[0](params, **spec[1])
            return spec(params)

        self.policy_opt = build(
            policy_optimizer,
            [self.policy[k] for k in nets.trainable_keys(self.policy)], policy_lr)
        self.vf_opt = build(
            vf_optimizer,
            [self.value[k] for k in nets.trainable_keys(self.value)], vf_lr)

    @staticmethod
    def _snapshot(params):
        return OrderedDict((k, v.clone().detach()) for k, v in params.items())

    # -- distributions -----------------------------------------------------
    def _dist(self, params, obs):
        if self.kind == 'gaussian':
            return nets.gaussian_dist(params, nets.POLICY_PREFIX, obs)
        return nets.categorical_dist(params, nets.POLICY_PREFIX, obs,
                                     self.double_softmax)

    def _log_prob(self, dist, actions):
        if self.kind == 'gaussian':
            return dist.log_prob(actions)
        return dist.log_prob(actions.long())

    def _entropy(self, obs):
        """``vpg.py:408-432``."""
        if self.stop_entropy_gradient:
            with torch.no_grad():
                ent = self._dist(self.policy, obs).entropy()
        else:
            ent = self._dist(self.policy, obs).entropy()
        if self.softplus_entropy:
            ent = F.softplus(ent)
        return ent

    def _objective(self, adv, obs, actions):
        new_ll = self._log_prob(self._dist(self.policy, obs), actions)
        if self.algo == 'vpg':
            return new_ll * adv  # vpg.py:452-454
        with torch.no_grad():
            old_ll = self._log_prob(self._dist(self.old_policy, obs), actions)
        ratio = (new_ll - old_ll).exp()
        clipped = torch.clamp(ratio, min=1 - self.clip, max=1 + self.clip)
        return torch.min(ratio * adv, clipped * adv)  # ppo.py:119-132

    def _policy_loss(self, obs, actions, adv):
        obj = self._objective(adv, obs, actions)
        if self.reg_entropy:
            obj = obj + self.ent_coeff * self._entropy(obs)
        return -obj.mean()

    def _kl(self, obs):
        """``vpg.py:381-406`` on whatever ``obs`` it is handed (padded)."""
        with torch.no_grad():
            old = self._dist(self.old_policy, obs)
        new = self._dist(self.policy, obs)
        return torch.distributions.kl.kl_divergence(old, new).mean()

    def _update_policy(self, obs_flat, actions_flat, adv_flat, used,
                       stream=None):
        """``vpg.py:244-245`` + ``_train_policy`` (``:250-272``)."""
        S = obs_flat.shape[0]
        if stream is None:
            stream = minibatch_index_stream(S, self.mb, self.epochs)
        for ids in stream:
            sel = slice(None) if ids is None else ids
            self.policy_opt.zero_grad()
            loss = self._policy_loss(obs_flat[sel], actions_flat[sel],
                                     adv_flat[sel])
            loss.backward()
            self.policy_opt.step()
            if used is not None and ids is not None:
                used.append(ids)

    # -- one iteration -----------------------------------------------------
    def train_once(self, batch, record_minibatches=False,
                   minibatch_ids=None):
        """``VPG._train_once`` on an :class:`oracle.batch.OracleEpisodeBatch`.

        Returns a dict with the 9 logged scalars (``vpg.py:186-199``), the
        intermediate tensors the parity tests compare, and (optionally) the
        minibatch id arrays in the order they were consumed.
        ``minibatch_ids = {'policy': [ids, ...], 'vf': [ids, ...]}`` replaces
        the ``BatchDataset`` streams by explicit id arrays (SURVEY.md section
        8e: the data-parallel tests feed the union of the ranks' minibatches).
        """
        obs = torch.Tensor(batch.padded_observations)
        rewards = torch.Tensor(batch.padded_rewards)
        returns = padded_returns(batch.padded_rewards, self.discount)
        valids = batch.lengths
        with torch.no_grad():
            baselines = nets.value_forward(self.value, obs)
        if self.max_entropy:
            rewards = rewards + self.ent_coeff * self._entropy(obs)

        obs_flat = torch.Tensor(batch.observations)
        actions_flat = torch.Tensor(batch.actions)
        returns_flat = torch.cat(filter_valids(returns, valids))
        adv_flat = vpg_compute_advantage(self.discount, self.gae_lambda,
                                         self.P, rewards, valids, baselines,
                                         self.center_adv, self.positive_adv)
        with torch.no_grad():
            pl_before = self._policy_loss(obs_flat, actions_flat, adv_flat)
            vl_before = nets.value_loss(self.value, obs_flat, returns_flat)
            kl_before = self._kl(obs)

        S = obs_flat.shape[0]
        used = {'policy': [], 'vf': []}
        # vpg.py:244-248 -- policy first, all epochs; then the value function.
        self._update_policy(obs_flat, actions_flat, adv_flat,
                            used['policy'] if record_minibatches else None,
                            None if minibatch_ids is None
                            else minibatch_ids['policy'])
        for ids in (minibatch_index_stream(S, self.mb, self.epochs)
                    if minibatch_ids is None else minibatch_ids['vf']):
            sel = slice(None) if ids is None else ids
            self.vf_opt.zero_grad()
            loss = nets.value_loss(self.value, obs_flat[sel],
                                   returns_flat[sel])
            loss.backward()
            self.vf_opt.step()
            if record_minibatches and ids is not None:
                used['vf'].append(ids)

        with torch.no_grad():
            pl_after = self._policy_loss(obs_flat, actions_flat, adv_flat)
            vl_after = nets.value_loss(self.value, obs_flat, returns_flat)
            kl_after = self._kl(obs)
            entropy = self._entropy(obs)

        self.old_policy = self._snapshot(self.policy)  # vpg.py:201
        stats, undiscounted = performance_stats(batch, self.discount)
        return {
            'policy/LossBefore': pl_before.item(),
            'policy/LossAfter': pl_after.item(),
            'policy/dLoss': (pl_before - pl_after).item(),
            'policy/KLBefore': kl_before.item(),
            'policy/KL': kl_after.item(),
            'policy/Entropy': entropy.mean().item(),
            'vf/LossBefore': vl_before.item(),
            'vf/LossAfter': vl_after.item(),
            'vf/dLoss': vl_before.item() - vl_after.item(),
            'returns_flat': returns_flat.numpy().copy(),
            'advantages_flat': adv_flat.detach().numpy().copy(),
            'baselines': baselines.numpy().copy(),
            'performance': stats,
            'average_return': float(np.mean(undiscounted)),
            'minibatches': used,
        }

    def state(self):
        pol = {k: v.detach().numpy().copy() for k, v in self.policy.items()}
        val = {k: v.detach().numpy().copy() for k, v in self.value.items()}
        return pol, val

    def adam_state(self, which='policy'):
        opt = self.policy_opt if which == 'policy' else self.vf_opt
        out = []
        for p in opt.param_groups[0]['params']:
            st = opt.state.get(p, {})
            if st:
                out.append((int(st['step']), st['exp_avg'].numpy().copy(),
                            st['exp_avg_sq'].numpy().copy()))
        return out


def clone_params(params):
    return copy.deepcopy(params)
