"""SURVEY.md section 8b, rows "Policy" and "Value function": the methods user code
and garage call on the plug-ins -- ``forward`` (a torch distribution + info dict),
``get_action(s)``, ``get_param_values`` / ``set_param_values``, ``state_dict`` key
names and shapes, ``compute_loss`` -- against outputs of the real
``GaussianMLPPolicy`` / ``GaussianMLPValueFunction`` (tests/golden/networks.npz,
four network sizes)."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(g, prefix):
    out = OrderedDict()
    for k in g.files:
        if k.startswith(prefix):
            out[k[len(prefix):]] = torch.from_numpy(np.asarray(g[k]))
    return out


@pytest.mark.parametrize('tag', ['tiny', 'c2', 'c3', 'deep'])
def test_policy_and_value_function_surface_matches_real_classes(golden, tag):
    from garage_amd._dtypes import Box, EnvSpec
    from garage_amd.policies import GaussianMLPPolicy, GaussianMLPValueFunction
    g = golden('networks')
    obs, act, ret = g[tag + '_obs'], g[tag + '_act'], g[tag + '_ret']
    hidden = tuple(int(v) for v in g[tag + '_hidden'])
    O, A = obs.shape[1], act.shape[1]
    spec = EnvSpec(Box(-np.inf, np.inf, (O, )), Box(-np.inf, np.inf, (A, )),
                   max_episode_length=10)
    ref_pol, ref_vf = _params(g, tag + '_pol:'), _params(g, tag + '_vf:')
    pol = GaussianMLPPolicy(spec, hidden_sizes=hidden)
    vf = GaussianMLPValueFunction(spec, hidden_sizes=hidden)
    # state_dict: the reference's key names and shapes, loadable as they are
    assert list(pol.state_dict()) == list(ref_pol)
    assert list(vf.state_dict()) == list(ref_vf)
    for k, v in pol.state_dict().items():
        assert tuple(v.shape) == tuple(ref_pol[k].shape), k
    pol.load_state_dict(ref_pol)
    vf.load_state_dict(ref_vf)
    assert pol.name == 'GaussianMLPPolicy'
    assert vf.name == 'GaussianMLPValueFunction'

    # forward: Independent(Normal) + info, as gaussian_mlp_policy.py:89-102
    dist, info = pol.forward(torch.from_numpy(obs))
    assert isinstance(dist, torch.distributions.Independent)
    assert np.allclose(info['mean'].cpu().numpy(), g[tag + '_mean'], atol=2e-6)
    assert np.allclose(info['log_std'].cpu().numpy(), g[tag + '_log_std'],
                       atol=1e-6)
    lp = dist.log_prob(torch.from_numpy(act).to(info['mean'].device))
    assert np.allclose(lp.cpu().numpy(), g[tag + '_log_prob'], atol=2e-5,
                       rtol=1e-5)
    assert np.allclose(dist.entropy().cpu().numpy(), g[tag + '_entropy'],
                       atol=1e-5)
    # (N, P, O) observations keep their leading dimensions
    stacked = torch.from_numpy(obs[:12]).reshape(3, 4, O)
    d2, i2 = pol(stacked)
    assert tuple(i2['mean'].shape) == (3, 4, A)
    assert np.allclose(i2['mean'].reshape(12, A).cpu().numpy(),
                       g[tag + '_mean'][:12], atol=2e-6)

    # get_actions / get_action: numpy in, numpy out, agent infos per action
    acts, infos = pol.get_actions(obs[:9])
    assert isinstance(acts, np.ndarray) and acts.shape == (9, A)
    assert acts.dtype == np.float32
    assert np.allclose(infos['mean'], g[tag + '_mean'][:9], atol=2e-6)
    assert np.allclose(infos['log_std'], g[tag + '_log_std'][:9], atol=1e-6)
    a1, i1 = pol.get_action(obs[0])
    assert a1.shape == (A, ) and i1['mean'].shape == (A, )
    assert np.allclose(i1['mean'], g[tag + '_mean'][0], atol=2e-6)
    pol.reset()
    pol.reset([True] * 9)

    # get_param_values / set_param_values round trip through another instance
    other = GaussianMLPPolicy(spec, hidden_sizes=hidden)
    other.set_param_values(pol.get_param_values())
    assert torch.equal(other.net.params, pol.net.params)

    # value function: forward flattens the last axis away, compute_loss is the NLL
    v = vf.forward(torch.from_numpy(obs))
    assert tuple(v.shape) == (obs.shape[0], )
    assert np.allclose(v.cpu().numpy(), g[tag + '_value'], atol=5e-6)
    v3 = vf(torch.from_numpy(obs[:12]).reshape(3, 4, O))
    assert tuple(v3.shape) == (3, 4)
    loss = vf.compute_loss(torch.from_numpy(obs), torch.from_numpy(ret))
    assert np.isclose(float(loss), float(g[tag + '_vf_loss']), rtol=1e-5,
                      atol=1e-6)


def test_nn_module_surface():
    """The ``nn.Module`` methods garage's own code and launchers call on a policy
    / value function (``torch/policies/policy.py:9-79``: ``parameters``,
    ``state_dict`` ...; ``.to()``, ``.train()`` / ``.eval()``, ``.zero_grad()``,
    ``.modules()``, calling the module): bookkeeping on the flat HBM buffers."""
    from garage_amd._dtypes import Box, EnvSpec
    from garage_amd.policies import GaussianMLPPolicy, GaussianMLPValueFunction
    spec = EnvSpec(Box(-np.inf, np.inf, (5, )), Box(-np.inf, np.inf, (2, )),
                   max_episode_length=10)
    torch.manual_seed(0)
    pol = GaussianMLPPolicy(spec, hidden_sizes=(8, 8))
    vf = GaussianMLPValueFunction(spec, hidden_sizes=(8, 8))
    for m in (pol, vf):
        assert m.train() is m and m.training is True
        assert m.eval() is m and m.training is False
        assert m.to('cuda') is m and m.to(torch.float32) is m and m.cuda() is m
        with pytest.raises(RuntimeError, match='no CPU fallback'):
            m.to('cpu')
        assert list(m.modules()) == [m] and list(m.children()) == []
        assert [n for n, _ in m.named_modules()] == ['']
        m.net.grads.fill_(1.0)
        m.zero_grad()
        assert float(m.net.grads.abs().sum()) == 0.0
        assert m.requires_grad_(False) is m and m.apply(lambda x: None) is m
        names = [n for n, _ in m.named_parameters()]
        # (the reference's attribute names: policy._module, value_function.module)
        assert names[0] == ('_module._init_std' if m is pol
                            else 'module._init_std')
        assert len(list(m.parameters())) == len(names)
    assert len(pol.buffers()) == 1 and len(vf.buffers()) == 0  # min_std_param
    obs = torch.randn(4, 5)
    dist, info = pol(obs)  # __call__ == forward
    dist2, _ = pol.forward(obs)
    assert torch.equal(dist.mean, dist2.mean) and 'mean' in info
    assert torch.equal(vf(obs), vf.forward(obs))
