"""The closed-form cases of the reference's own policy / value-function tests
(``tests/garage/torch/policies/test_gaussian_mlp_policy.py:21-200``,
``tests/garage/torch/value_functions/test_gaussian_mlp_value_function.py``): linear
networks (``hidden_nonlinearity=None``) with all-ones weights, so that every mean is
``obs_dim * prod(hidden_sizes)`` and the variance ``init_std ** 2``."""
import pickle

import numpy as np
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu

HIDDEN = [(1, ), (2, ), (3, ), (1, 4), (3, 5)]


def _spec(obs_dim=4, act_dim=2):
    from garage_amd._dtypes import Box, EnvSpec
    return EnvSpec(Box(-1.0, 1.0, (obs_dim, )), Box(-1.0, 1.0, (act_dim, )),
                   max_episode_length=10)


def _policy(spec, hidden_sizes, init_std=2.0):
    from garage_amd.policies import GaussianMLPPolicy
    return GaussianMLPPolicy(env_spec=spec, hidden_sizes=hidden_sizes,
                             init_std=init_std, hidden_nonlinearity=None,
                             std_parameterization='exp',
                             hidden_w_init=nn.init.ones_,
                             output_w_init=nn.init.ones_)


@pytest.mark.parametrize('as_numpy', [False, True])
@pytest.mark.parametrize('hidden_sizes', HIDDEN)
def test_get_action(hidden_sizes, as_numpy):
    """``test_get_action`` / ``test_get_action_np``."""
    spec = _spec()
    obs_dim, act_dim = 4, 2
    obs = np.ones(obs_dim, np.float32) if as_numpy else torch.ones(obs_dim)
    policy = _policy(spec, hidden_sizes)
    dist = policy(torch.ones(obs_dim))[0]
    expected_mean = np.full((act_dim, ), obs_dim * float(np.prod(hidden_sizes)),
                            np.float32)
    action, prob = policy.get_action(obs)
    assert np.array_equal(np.asarray(prob['mean']), expected_mean)
    assert torch.equal(dist.variance.cpu(),
                       torch.full((act_dim, ), 4.0, dtype=torch.float))
    assert np.asarray(action).shape == (act_dim, )


@pytest.mark.parametrize('as_numpy', [False, True])
@pytest.mark.parametrize('batch_size, hidden_sizes',
                         [(1, (1, )), (4, (3, )), (10, (2, 4)), (5, (3, 5))])
def test_get_actions(batch_size, hidden_sizes, as_numpy):
    """``test_get_actions`` / ``test_get_actions_np``."""
    spec = _spec()
    obs_dim, act_dim = 4, 2
    obs = (np.ones((batch_size, obs_dim), np.float32) if as_numpy else
           torch.ones(batch_size, obs_dim))
    policy = _policy(spec, hidden_sizes)
    dist = policy(torch.ones(batch_size, obs_dim))[0]
    expected_mean = np.full((batch_size, act_dim),
                            obs_dim * float(np.prod(hidden_sizes)), np.float32)
    action, prob = policy.get_actions(obs)
    assert np.array_equal(np.asarray(prob['mean']), expected_mean)
    assert torch.equal(dist.variance.cpu(),
                       torch.full((batch_size, act_dim), 4.0, dtype=torch.float))
    assert np.asarray(action).shape == (batch_size, act_dim)


@pytest.mark.parametrize('batch_size, hidden_sizes',
                         [(1, (1, )), (4, (3, )), (10, (2, 4))])
def test_is_pickleable(batch_size, hidden_sizes):
    """``test_is_pickleable``: the pickled policy gives the same means."""
    spec = _spec()
    obs = torch.ones(batch_size, 4)
    policy = _policy(spec, hidden_sizes)
    _, prob1 = policy.get_actions(obs)
    policy2 = pickle.loads(pickle.dumps(policy))
    action2, prob2 = policy2.get_actions(obs)
    assert np.array_equal(np.asarray(prob1['mean']), np.asarray(prob2['mean']))
    assert np.asarray(action2).shape == (batch_size, 2)


@pytest.mark.parametrize('hidden_sizes', [(1, ), (2, ), (3, 5)])
def test_value_function_closed_form(hidden_sizes):
    """A linear value network with all-ones weights returns
    ``obs_dim * prod(hidden_sizes)`` per state (the construction of
    ``test_gaussian_mlp_value_function.py``); its NLL against that target with unit
    std is ``0.5 log(2 pi)``."""
    import math

    from garage_amd.policies import GaussianMLPValueFunction
    spec = _spec()
    vf = GaussianMLPValueFunction(env_spec=spec, hidden_sizes=hidden_sizes,
                                  hidden_nonlinearity=None,
                                  hidden_w_init=nn.init.ones_,
                                  output_w_init=nn.init.ones_)
    obs = torch.ones(6, 4)
    want = 4.0 * float(np.prod(hidden_sizes))
    out = vf.forward(obs).cpu().numpy().reshape(-1)
    assert np.array_equal(out, np.full(6, want, np.float32))
    loss = vf.compute_loss(obs, torch.full((6, ), want))
    assert math.isclose(float(loss), 0.5 * math.log(2 * math.pi), rel_tol=1e-6)
