"""Oracle: MLP / Gaussian / categorical heads as plain functions of a param dict.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Restates
  * ``torch/modules/multi_headed_mlp_module.py:136-151`` + ``mlp_module.py:62-73``
    (Linear -> hidden_nonlinearity stack -- tanh by default -- linear output head)
  * ``torch/modules/gaussian_mlp_module.py:158-192,288-305`` (scalar log-std
    broadcast, lower clamp at log(min_std), exp parameterisation,
    ``Independent(Normal(mean, std), 1)``)
  * ``torch/policies/gaussian_mlp_policy.py:89-102`` (agent_info mean/log_std)
  * ``torch/value_functions/gaussian_mlp_value_function.py:81-112`` (value =
    distribution mean, loss = Gaussian NLL with a learned scalar log-std)
  * ``torch/policies/categorical_cnn_policy.py:138-139`` for the categorical
    head convention (softmax output passed as ``logits=``).

Parameters live in an ordered ``dict[str, torch.Tensor]`` that uses the
reference's ``state_dict`` key names, so a golden ``state_dict`` captured from
the real classes drops straight in.
"""
import contextlib
import math
from collections import OrderedDict

import numpy as np
import torch
from torch.distributions import Categorical, Independent, Normal

POLICY_PREFIX = '_module.'
VALUE_PREFIX = 'module.'


def xavier_uniform(rng, fan_out, fan_in):
    """``nn.init.xavier_uniform_`` bound, drawn from a numpy RandomState."""
    bound = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-bound, bound, size=(fan_out, fan_in)).astype(np.float32)


def init_gaussian_mlp(rng, prefix, in_dim, out_dim, hidden_sizes,
                      init_std=1.0, min_std=None):
    """Param dict shaped like ``GaussianMLPModule.state_dict()``.

    xavier-uniform weights and zero biases are the reference defaults
    (``gaussian_mlp_policy.py:55-60``); values come from ``rng`` so product and
    oracle can be given identical parameters.
    """
    p = OrderedDict()
    p[prefix + '_init_std'] = torch.tensor([math.log(init_std)],
                                           dtype=torch.float32)
    if min_std is not None:
        p[prefix + 'min_std_param'] = torch.tensor([min_std]).log()
    prev = in_dim
    for i, h in enumerate(hidden_sizes):
        base = '{}_mean_module._layers.{}.linear.'.format(prefix, i)
        p[base + 'weight'] = torch.from_numpy(xavier_uniform(rng, h, prev))
        p[base + 'bias'] = torch.zeros(h)
        prev = h
    base = prefix + '_mean_module._output_layers.0.linear.'
    p[base + 'weight'] = torch.from_numpy(xavier_uniform(rng, out_dim, prev))
    p[base + 'bias'] = torch.zeros(out_dim)
    return p


def trainable_keys(params):
    """Everything but the registered buffers (``min_std_param``,
    ``max_std_param`` and, with ``learn_std=False``, ``init_std``)."""
    return [k for k in params
            if not k.endswith(('min_std_param', 'max_std_param', '.init_std'))]


def n_hidden(params, prefix):
    i = 0
    while '{}_mean_module._layers.{}.linear.weight'.format(prefix, i) in params:
        i += 1
    return i


# ``hidden_nonlinearity`` (mlp_module.py:43-44) is a constructor argument of the
# reference modules, not part of their state_dict: the parameter dicts cannot carry
# it, so it is set around a computation, per network (None = linear).
_HIDDEN = [{POLICY_PREFIX: torch.tanh, VALUE_PREFIX: torch.tanh}]
_OUTPUT = [{POLICY_PREFIX: None, VALUE_PREFIX: None}]


@contextlib.contextmanager
def hidden_nonlinearity(policy=torch.tanh, value=torch.tanh):
    _HIDDEN.append({POLICY_PREFIX: policy, VALUE_PREFIX: value})
    try:
        yield
    finally:
        _HIDDEN.pop()


@contextlib.contextmanager
def output_nonlinearity(policy=None, value=None):
    """``output_nonlinearity`` of the mean / value MLP (mlp_module.py:52-53)."""
    _OUTPUT.append({POLICY_PREFIX: policy, VALUE_PREFIX: value})
    try:
        yield
    finally:
        _OUTPUT.pop()


# ``std_parameterization`` of the policy likewise ('exp' or 'softplus',
# gaussian_mlp_module.py:178-181)
_STD_PARAM = ['exp']


@contextlib.contextmanager
def std_parameterization(kind):
    _STD_PARAM.append(kind)
    try:
        yield
    finally:
        _STD_PARAM.pop()


def mlp_mean(params, prefix, x):
    """MLP trunk (tanh unless ``hidden_nonlinearity`` says otherwise) + linear head."""
    act = _HIDDEN[-1].get(prefix, torch.tanh)
    for i in range(n_hidden(params, prefix)):
        base = '{}_mean_module._layers.{}.linear.'.format(prefix, i)
        ln = '{}_mean_module._layers.{}.layer_normalization.'.format(prefix, i)
        if ln + 'weight' in params:  # layer_normalization=True
            # (multi_headed_mlp_module.py:77-81: nn.LayerNorm(prev_size) first)
            x = torch.nn.functional.layer_norm(x, (x.shape[-1], ),
                                               params[ln + 'weight'],
                                               params[ln + 'bias'], 1e-5)
        x = torch.nn.functional.linear(x, params[base + 'weight'],
                                       params[base + 'bias'])
        if act is not None:
            x = act(x)
    base = prefix + '_mean_module._output_layers.0.linear.'
    x = torch.nn.functional.linear(x, params[base + 'weight'], params[base + 'bias'])
    out_act = _OUTPUT[-1].get(prefix)
    return x if out_act is None else out_act(x)


def gaussian_dist(params, prefix, x):
    """``GaussianMLPBaseModule.forward`` (``gaussian_mlp_module.py:158-192``)."""
    mean = mlp_mean(params, prefix, x)
    # learn_std=False registers the log-std as the buffer ``init_std`` instead of
    # the parameter ``_init_std`` (gaussian_mlp_module.py:124-130)
    std_key = prefix + ('_init_std' if prefix + '_init_std' in params
                        else 'init_std')
    log_std = torch.zeros(*mean.shape) + params[std_key]
    lo, hi = prefix + 'min_std_param', prefix + 'max_std_param'
    if lo in params or hi in params:  # gaussian_mlp_module.py:171-176
        log_std = log_std.clamp(
            min=params[lo].item() if lo in params else None,
            max=params[hi].item() if hi in params else None)
    if _STD_PARAM[-1] == 'softplus' and prefix == POLICY_PREFIX:
        std = log_std.exp().exp().add(1.).log()  # gaussian_mlp_module.py:181
    else:
        std = log_std.exp()
    return Independent(Normal(mean, std), 1)


def policy_forward(params, obs):
    """(dist, agent_info) of ``GaussianMLPPolicy.forward``."""
    dist = gaussian_dist(params, POLICY_PREFIX, obs)
    return dist, dict(mean=dist.mean, log_std=(dist.variance**.5).log())


def value_forward(params, obs):
    """``GaussianMLPValueFunction.forward``: ``(..., O) -> (...)``."""
    return gaussian_dist(params, VALUE_PREFIX, obs).mean.flatten(-2)


def value_loss(params, obs, returns):
    """``GaussianMLPValueFunction.compute_loss``: ``-mean log N(G | v, s^2)``."""
    dist = gaussian_dist(params, VALUE_PREFIX, obs)
    return -dist.log_prob(returns.reshape(-1, 1)).mean()


def categorical_dist(params, prefix, x, double_softmax=True):
    """Categorical head for the discrete configs (no torch MLP precedent).

    The only torch categorical policies in the reference feed
    ``softmax(net(x))`` to ``Categorical(logits=...)``
    (``categorical_cnn_policy.py:138-139``); ``double_softmax=True`` keeps that
    convention, ``False`` treats the MLP output as logits.  Pinned by the
    real ``CategoricalCNNPolicy`` configured as an MLP
    (``tests/golden/train_once_categorical.npz``,
    ``tests/test_oracle_golden.py::test_golden_categorical_train_once``).
    """
    out = mlp_mean(params, prefix, x)
    if double_softmax:
        out = torch.softmax(out, dim=-1)
    return Categorical(logits=out)
