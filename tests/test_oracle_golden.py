"""The oracle against (a) the reference's own literal test vectors and
(b) golden outputs captured from the real reference (tests/golden/*.npz).

CPU only.  These tests are what "pins" the oracle (oracle/__init__.py).
"""
import math

import numpy as np
import pytest
import torch

from oracle import batch as ob
from oracle import envs as oenvs
from oracle import networks as nets
from oracle import returns as orr
from oracle import sampler as osamp
from oracle.ppo import OraclePPO


# -- reference literal vectors ------------------------------------------------
def test_log_performance_literals():
    """tests/garage/test_functions.py:49-97 (the only pin of discount_cumsum)."""
    rewards = np.array([
        0.34026529, 0.58263177, 0.84307509, 0.97651095, 0.81723901,
        0.22631398, 0.03421301, 0.97515046, 0.64311832, 0.65068933,
        0.17657714, 0.04783857, 0.73904013, 0.41364329, 0.52235551,
        0.24203526, 0.43328910
    ])
    lengths = np.array([10, 5, 1, 1])
    st = ([0] + [1] * 8 + [2] + [0] + [1] * 3 + [2] + [0] + [0])
    success = np.array([0, 0, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 1, 0, 1],
                       dtype=bool)
    b = ob.OracleEpisodeBatch(observations=np.ones((17, 3), np.float32),
                              last_observations=np.ones((4, 3), np.float32),
                              actions=np.zeros((17, 2), np.float32),
                              rewards=rewards, step_types=np.asarray(st),
                              lengths=lengths, env_infos={'success': success})
    stats, _ = ob.performance_stats(b, 0.8)
    assert stats['NumEpisodes'] == 4
    assert math.isclose(stats['SuccessRate'], 0.75)
    assert math.isclose(stats['TerminationRate'], 0.5)
    assert math.isclose(stats['AverageDiscountedReturn'], 1.1131040640673113)
    assert math.isclose(stats['AverageReturn'], 2.1659965525)
    assert math.isclose(stats['StdReturn'], 2.354067152038576)


MT_LITERAL = dict(
    lengths=np.array([10, 5, 1, 1]),
    rewards=np.array([
        0.34026529, 0.58263177, 0.84307509, 0.97651095, 0.81723901,
        0.22631398, 0.03421301, 0.97515046, 0.64311832, 0.65068933,
        0.17657714, 0.04783857, 0.73904013, 0.41364329, 0.52235551,
        0.24203526, 0.43328910
    ]),
    success=np.array([0, 0, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 1, 0, 1],
                     dtype=bool),
    task_name=np.array(['env1'] * 10 + ['env2'] * 5 + ['env1'] + ['env3']),
    task_id=np.array([1] * 10 + [3] * 5 + [1] + [4]),
    name_map={1: 'env1', 3: 'env2', 4: 'env3', 5: 'env4'},
)


def check_multitask_literals(rec, by):
    """The assertions of tests/garage/test_functions.py:101-200."""
    names = ['env1', 'env2', 'env3'] + (['env4'] if by == 'task_id' else [])
    for name in names:
        assert rec[name + '/Iteration'] == 7
    assert rec['env1/NumEpisodes'] == 2
    assert rec['env2/NumEpisodes'] == 1
    assert rec['env3/NumEpisodes'] == 1
    assert math.isclose(rec['env1/SuccessRate'], 0.5)
    assert math.isclose(rec['env2/SuccessRate'], 1.0)
    assert math.isclose(rec['env3/SuccessRate'], 1.0)
    if by == 'task_id':
        assert rec['env4/NumEpisodes'] == 0
        assert math.isnan(rec['env4/SuccessRate'])
        assert math.isnan(rec['env4/AverageReturn'])


@pytest.mark.parametrize('by', ['task_name', 'task_id'])
def test_log_multitask_performance_literals(by):
    """tests/garage/test_functions.py:101-200 (both reference tests)."""
    lit = MT_LITERAL
    S = int(lit['lengths'].sum())
    b = ob.OracleEpisodeBatch(
        observations=np.ones((S, 3), np.float32),
        last_observations=np.ones((4, 3), np.float32),
        actions=np.zeros((S, 2), np.float32), rewards=lit['rewards'],
        step_types=np.ones(S, dtype=np.int64), lengths=lit['lengths'],
        env_infos={'success': lit['success'], by: lit[by]})
    rec, _ = ob.multitask_performance_stats(
        7, b, 0.8, name_map=lit['name_map'] if by == 'task_id' else None)
    check_multitask_literals(rec, by)


@pytest.mark.parametrize('discount', [1, 0.95])
@pytest.mark.parametrize('num_eps', [1, 5])
@pytest.mark.parametrize('gae_lambda', [0, 0.5, 1])
@pytest.mark.parametrize('which', [0, 1, 2])
def test_compute_advantages_reference_cases(discount, num_eps, gae_lambda,
                                            which):
    """tests/garage/torch/test_functions.py:86-117, same inline recursion."""
    r, b = [(np.ones(6), np.zeros(6)),
            (np.array([3, 1, 4, 1, 5, 9]), np.arange(6)),
            (np.ones(6), np.array([1, 1, 2, 3, 5, 8]))][which]
    rewards = torch.Tensor(np.repeat(r[None], num_eps, 0))
    base = torch.Tensor(np.repeat(b[None], num_eps, 0))
    expected = torch.zeros(rewards.shape)
    for i in range(num_eps):
        acc = 0
        for j in range(6):
            acc = acc * discount * gae_lambda
            acc += rewards[i][-j - 1] - base[i][-j - 1]
            acc += discount * base[i][-j] if j else 0
            expected[i][-j - 1] = acc
    got = orr.compute_advantages(discount, gae_lambda, 6, base, rewards)
    assert torch.allclose(expected, got)
    assert np.allclose(orr.gae_padded_f64(discount, gae_lambda, base, rewards),
                       expected.numpy(), atol=1e-5)


def test_pad_batch_array_reference_case():
    """tests/garage/np/test_functions.py:81-88."""
    lens = [1, 2, 3, 4]
    arr = np.arange(10)
    out = orr.pad_batch_array(arr, lens)
    assert out.shape == (4, 4)
    assert (out[1] == [1, 2, 0, 0]).all() and (out[3] == [6, 7, 8, 9]).all()


def test_step_type_truth_table_reference():
    """tests/garage/test_dtypes.py:290-318."""
    g = ob.StepType.get_step_type
    assert g(1, 5, False) == ob.StepType.FIRST
    assert g(2, 5, False) == ob.StepType.MID
    assert g(2, None, False) == ob.StepType.MID
    assert g(5, 5, False) == ob.StepType.TIMEOUT
    assert g(5, 5, True) == ob.StepType.TIMEOUT
    assert g(1, 5, True) == ob.StepType.TERMINAL
    with pytest.raises(ValueError):
        g(0, 5, False)


def test_gaussian_module_closed_form_reference():
    """tests/garage/torch/modules/test_gaussian_mlp_module.py:98-123:
    all-ones weights, linear activations -> mean = in_dim * prod(hidden)."""
    for in_dim, out_dim, hs in ((5, 1, (1, )), (5, 2, (2, 2)), (5, 1, (3, 3))):
        p = {}
        prev = in_dim
        for i, h in enumerate(hs):
            p['m._mean_module._layers.%d.linear.weight' % i] = torch.ones(
                h, prev)
            p['m._mean_module._layers.%d.linear.bias' % i] = torch.zeros(h)
            prev = h
        p['m._mean_module._output_layers.0.linear.weight'] = torch.ones(
            out_dim, prev)
        p['m._mean_module._output_layers.0.linear.bias'] = torch.zeros(out_dim)
        x = torch.ones(in_dim) * 1e-3  # tanh ~ identity to 1e-6 relative
        mean = nets.mlp_mean(p, 'm.', x)
        expect = 1e-3 * in_dim * float(np.prod(hs))
        assert torch.allclose(mean, torch.full((out_dim, ), expect), rtol=1e-4)


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    r = oenvs.philox4x32(0, 0, 0, 0, 0)
    assert [int(v) for v in r] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c,
                                   0x9b00dbd8]
    r = oenvs.philox4x32(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff,
                         0xffffffffffffffff)
    assert [int(v) for v in r] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6,
                                   0x6d5451fd]
    r = oenvs.philox4x32(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344,
                         (0x299f31d0 << 32) | 0xa4093822)
    assert [int(v) for v in r] == [0xd16cfe09, 0x94fdcceb, 0x5001e420,
                                   0x24126ea1]


# -- goldens captured from the real reference ----------------------------------
def test_golden_discount_cumsum(golden):
    g = golden('returns')
    start, firsts = 0, []
    for L in g['lp_lengths']:
        firsts.append(orr.discount_cumsum(g['lp_rewards'][start:start + L],
                                          0.8)[0])
        start += L
    assert np.array_equal(np.asarray(firsts), g['lp_first_returns'])
    for i in range(3):
        gamma = float(g['dc_g_%d' % i])
        for kind in ('64', '32'):
            x, y = g['dc_x%s_%d' % (kind, i)], g['dc_y%s_%d' % (kind, i)]
            assert np.array_equal(orr.discount_cumsum(x, gamma), y)
            assert np.allclose(orr.discount_cumsum_recursive(x, gamma), y,
                               rtol=1e-12, atol=1e-12)


def test_golden_compute_advantages(golden):
    g = golden('advantages')
    for i in range(int(g['n_test_cases'])):
        rew, base = g['t%d_in' % i]
        d, lam, P = g['t%d_cfg' % i]
        got = orr.compute_advantages(d, lam, int(P), torch.Tensor(base),
                                     torch.Tensor(rew))
        assert np.allclose(got.numpy(), g['t%d_adv' % i], atol=1e-6)
    for k in range(int(g['n_ragged_cases'])):
        rew, base, lens = (g['r%d_rewards' % k], g['r%d_base' % k],
                           g['r%d_lens' % k])
        d, lam, P, v0 = g['r%d_cfg' % k]
        P = int(P)
        ref = g['r%d_adv' % k]
        got = orr.compute_advantages(d, lam, P, torch.Tensor(base),
                                     torch.Tensor(rew)).numpy()
        assert np.allclose(got, ref, atol=2e-6)
        assert np.allclose(orr.gae_padded_f64(d, lam, base, rew), ref,
                           atol=1e-5)
        # the closed form the HIP scan uses (no padding touched)
        vals = np.concatenate([base[i, :L] for i, L in enumerate(lens)])
        rews = np.concatenate([rew[i, :L] for i, L in enumerate(lens)])
        packed = orr.gae_ragged_closed_form_f64(d, lam, P, vals, rews, lens,
                                                float(v0))
        ref_packed = np.concatenate([ref[i, :L] for i, L in enumerate(lens)])
        assert np.allclose(packed, ref_packed, atol=1e-5)


def test_golden_vpg_compute_advantage(golden):
    g = golden('compute_advantage')
    for k in range(int(g['n_cases'])):
        center, positive = g['c%d_flags' % k]
        got = orr.vpg_compute_advantage(0.99, 0.97, 8,
                                        torch.Tensor(g['c%d_rewards' % k]),
                                        g['c%d_lens' % k],
                                        torch.Tensor(g['c%d_base' % k]),
                                        bool(center), bool(positive)).numpy()
        assert np.allclose(got, g['c%d_adv' % k], atol=1e-5, equal_nan=True)


def test_golden_padding_steptypes(golden):
    g = golden('padding_steptypes')
    assert np.array_equal(orr.pad_batch_array(g['obs'], g['lens'], 100),
                          g['padded_obs'])
    assert np.array_equal(orr.pad_batch_array(g['rew'], g['lens'], 100),
                          g['padded_rew'])
    assert np.array_equal(orr.pad_batch_array(g['obs'], g['lens']),
                          g['padded_obs_default'])
    for step_cnt, max_len, done, expect in g['steptype_table']:
        ml = None if max_len < 0 else int(max_len)
        assert int(ob.StepType.get_step_type(int(step_cnt), ml,
                                             bool(done))) == expect


class _Scripted:
    """Same scripted agent as tests/golden/make_golden.py."""

    def __init__(self, act_dim):
        self.act_dim, self.calls = act_dim, 0

    def reset(self, do_resets=None):
        pass

    def get_actions(self, observations):
        obs = np.asarray(observations, dtype=np.float32)
        a = np.zeros((obs.shape[0], self.act_dim), dtype=np.float32)
        a[:, 0] = obs.sum(axis=1)
        a[:, -1] = self.calls
        self.calls += 1
        return a, {'tag': a[:, 0] * 2}

    def get_action(self, observation):
        a, info = self.get_actions(np.asarray(observation)[None])
        return a[0], {k: v[0] for k, v in info.items()}


def _check_batch(g, prefix, eps, check_obs=True):
    if check_obs:
        assert np.array_equal(eps.observations, g[prefix + 'observations'])
    assert np.array_equal(eps.last_observations,
                          g[prefix + 'last_observations'])
    assert np.array_equal(eps.actions, g[prefix + 'actions'])
    assert np.array_equal(eps.rewards, g[prefix + 'rewards'])
    assert np.array_equal([int(s) for s in eps.step_types],
                          g[prefix + 'step_types'])
    assert np.array_equal(eps.lengths, g[prefix + 'lengths'])
    assert str(eps.lengths.dtype) == str(g[prefix + 'lengths_dtype'])
    assert str(eps.rewards.dtype) == str(g[prefix + 'rewards_dtype'])
    assert np.array_equal(eps.agent_infos['tag'], g[prefix + 'agent_tag'])


def test_golden_sampler_bookkeeping(golden):
    g = golden('sampler')
    P, n = [int(v) for v in g['cfg']]
    cyc = g['cycles']

    def envs():
        return [oenvs.CountingEnv(i, cyc[i], P) for i in range(n)]

    # VecWorker, with the aliasing bug switched on -> equal field by field
    s = osamp.OracleLocalSampler(_Scripted(2), [envs()], max_episode_length=P,
                                 n_workers=1,
                                 worker_class=osamp.OracleVecWorker,
                                 worker_args=dict(n_envs=n, alias_bug=True))
    _check_batch(g, 'vec_', s.obtain_samples(0, 30, None))
    _check_batch(g, 'vec2_', s.obtain_samples(1, 17, None))
    assert s.total_env_steps == int(g['vec_total_env_steps'])
    # ... and off: everything but the observations is unchanged, and the
    # observations are the ones each env really showed the agent.
    s = osamp.OracleLocalSampler(_Scripted(2), [envs()], max_episode_length=P,
                                 n_workers=1,
                                 worker_class=osamp.OracleVecWorker,
                                 worker_args=dict(n_envs=n))
    eps = s.obtain_samples(0, 30, None)
    _check_batch(g, 'vec_', eps, check_obs=False)
    assert not np.array_equal(eps.observations, g['vec_observations'])
    start = 0
    for L in eps.lengths:
        ep_obs = eps.observations[start:start + L]
        assert np.array_equal(ep_obs[:, 2], np.arange(L))  # t = 0..L-1
        assert len(set(ep_obs[:, 0])) == 1
        start += L
    # DefaultWorker
    s = osamp.OracleLocalSampler(_Scripted(2), envs(), max_episode_length=P,
                                 n_workers=n,
                                 worker_class=osamp.OracleDefaultWorker)
    _check_batch(g, 'def_', s.obtain_exact_episodes(3, None))
    # FragmentWorker
    for tpc in (1, 2):
        s = osamp.OracleLocalSampler(
            _Scripted(2), [envs()], max_episode_length=P, n_workers=1,
            worker_class=osamp.OracleFragmentWorker,
            worker_args=dict(n_envs=n, timesteps_per_call=tpc))
        _check_batch(g, 'frag%d_' % tpc, s.obtain_samples(0, 20, None))


def _params(g, prefix):
    from collections import OrderedDict
    out = OrderedDict()
    for k in g.files:
        if k.startswith(prefix):
            out[k[len(prefix):]] = torch.from_numpy(g[k].copy())
    return out


def test_golden_networks(golden):
    g = golden('networks')
    for tag in ('tiny', 'c2', 'c3', 'deep'):
        pol, vf = _params(g, tag + '_pol:'), _params(g, tag + '_vf:')
        obs, act, ret = (torch.from_numpy(g[tag + '_' + k])
                         for k in ('obs', 'act', 'ret'))
        with torch.no_grad():
            dist, info = nets.policy_forward(pol, obs)
            assert np.allclose(info['mean'], g[tag + '_mean'], atol=1e-6)
            assert np.allclose(info['log_std'], g[tag + '_log_std'], atol=1e-6)
            assert np.allclose(dist.log_prob(act), g[tag + '_log_prob'],
                               atol=1e-5, rtol=1e-6)
            assert np.allclose(dist.entropy(), g[tag + '_entropy'], atol=1e-6)
            assert np.allclose(nets.value_forward(vf, obs), g[tag + '_value'],
                               atol=1e-6)
            assert np.allclose(nets.value_loss(vf, obs, ret),
                               g[tag + '_vf_loss'], atol=1e-6)


TRAIN_CASES = {
    'ppo': dict(algo='ppo'),
    'ppo_pos': dict(algo='ppo', positive_adv=True),
    'ppo_reg': dict(algo='ppo', entropy_method='regularized',
                    policy_ent_coeff=0.02),
    'ppo_max': dict(algo='ppo', entropy_method='max', policy_ent_coeff=0.05,
                    center_adv=False, stop_entropy_gradient=True,
                    use_softplus_entropy=True),
    'vpg': dict(algo='vpg', gae_lambda=1),
    'ppo_full': dict(algo='ppo'),
}

LOG_KEYS = {
    'policy/LossBefore': 'GaussianMLPPolicy/LossBefore',
    'policy/LossAfter': 'GaussianMLPPolicy/LossAfter',
    'policy/dLoss': 'GaussianMLPPolicy/dLoss',
    'policy/KLBefore': 'GaussianMLPPolicy/KLBefore',
    'policy/KL': 'GaussianMLPPolicy/KL',
    'policy/Entropy': 'GaussianMLPPolicy/Entropy',
    'vf/LossBefore': 'GaussianMLPValueFunction/LossBefore',
    'vf/LossAfter': 'GaussianMLPValueFunction/LossAfter',
    'vf/dLoss': 'GaussianMLPValueFunction/dLoss',
}


@pytest.mark.parametrize('tag', sorted(TRAIN_CASES))
def test_golden_train_once(golden, tag):
    """Two consecutive real ``_train_once`` iterations."""
    g = golden('train_once')
    O, A, P, E, mb = [int(v) for v in g[tag + '_cfg']]
    kw = dict(TRAIN_CASES[tag])
    if kw['algo'] == 'vpg':
        # VPG's own defaults (vpg.py:56-105): lambda 1, lr = Adam default.
        kw.setdefault('gae_lambda', 1)
    algo = OraclePPO(_params(g, tag + '_pol0:'), _params(g, tag + '_vf0:'),
                     max_episode_length=P, max_optimization_epochs=E,
                     minibatch_size=None if mb < 0 else mb, **kw)
    for it in range(2):
        pre = '%s_it%d_' % (tag, it)
        lens = g[pre + 'lengths']
        b = ob.OracleEpisodeBatch(
            observations=g[pre + 'observations'],
            last_observations=np.zeros((len(lens), O), np.float32),
            actions=g[pre + 'actions'], rewards=g[pre + 'rewards'],
            step_types=g[pre + 'step_types'], lengths=lens,
            max_episode_length=P)
        np.random.seed(int(g[pre + 'np_seed']))
        out = algo.train_once(b)
        for mine, theirs in LOG_KEYS.items():
            assert np.isclose(out[mine], float(g[pre + 'log:' + theirs]),
                              atol=1e-5, rtol=1e-5), (mine, it)
        assert np.isclose(out['average_return'], float(g[pre + 'avg_return']))
        perf = out['performance']
        for k in ('AverageDiscountedReturn', 'AverageReturn', 'StdReturn',
                  'MaxReturn', 'MinReturn', 'TerminationRate', 'NumEpisodes'):
            assert np.isclose(perf[k], float(g[pre + 'log:Evaluation/' + k]))
        pol, vf = algo.state()
        for k, v in pol.items():
            assert np.allclose(v, g[pre + 'pol:' + k], atol=1e-6), k
        for k, v in vf.items():
            assert np.allclose(v, g[pre + 'vf:' + k], atol=1e-6), k
        for which, name in (('policy', 'pol'), ('vf', 'vf')):
            for j, (step, m, v) in enumerate(algo.adam_state(which)):
                assert step == int(g['%sadam_%s_%d_step' % (pre, name, j)])
                assert np.allclose(m, g['%sadam_%s_%d_m' % (pre, name, j)],
                                   atol=1e-7)
                assert np.allclose(v, g['%sadam_%s_%d_v' % (pre, name, j)],
                                   atol=1e-9)


@pytest.mark.parametrize('tag', sorted(__import__('_categorical_golden').CASES))
def test_golden_categorical_train_once(golden, tag):
    """The categorical head against the REAL reference: two ``_train_once``
    iterations of the real PPO / VPG on the real ``CategoricalCNNPolicy``
    configured as an MLP (tests/_categorical_golden.py), and its forward
    distribution at fixed inputs."""
    import _categorical_golden as cg
    g = golden('train_once_categorical')
    O, n_act, P, E, mb = [int(v) for v in g[tag + '_cfg']]
    kw = dict(cg.CASES[tag])
    kw['algo'] = 'vpg' if tag == 'vpg' else 'ppo'
    if kw['algo'] == 'vpg':
        kw.setdefault('gae_lambda', 1)
    pol0 = cg.policy_params(g, tag + '_pol0:')
    with torch.no_grad():
        dist = nets.categorical_dist(pol0, nets.POLICY_PREFIX,
                                     torch.from_numpy(g[tag + '_fwd_obs']))
        act = torch.from_numpy(g[tag + '_fwd_act'])
        assert np.allclose(dist.probs, g[tag + '_fwd_probs'], atol=1e-6)
        assert np.allclose(dist.log_prob(act), g[tag + '_fwd_log_prob'],
                           atol=1e-6)
        assert np.allclose(dist.entropy(), g[tag + '_fwd_entropy'], atol=1e-6)
    algo = OraclePPO(pol0, cg.value_params(g, tag + '_vf0:'),
                     max_episode_length=P, policy_kind='categorical',
                     max_optimization_epochs=E,
                     minibatch_size=None if mb < 0 else mb,
                     policy_lr=1e-3, vf_lr=1e-3, **kw)
    for it in range(2):
        pre = '%s_it%d_' % (tag, it)
        lens = g[pre + 'lengths']
        b = ob.OracleEpisodeBatch(
            observations=g[pre + 'observations'],
            last_observations=np.zeros((len(lens), O), np.float32),
            actions=g[pre + 'actions'], rewards=g[pre + 'rewards'],
            step_types=g[pre + 'step_types'], lengths=lens,
            max_episode_length=P)
        np.random.seed(int(g[pre + 'np_seed']))
        out = algo.train_once(b)
        for mine, theirs in cg.LOG_KEYS.items():
            assert np.isclose(out[mine], float(g[pre + 'log:' + theirs]),
                              atol=1e-5, rtol=1e-5), (mine, it)
        assert np.isclose(out['average_return'], float(g[pre + 'avg_return']))
        pol, vf = algo.state()
        want = cg.policy_params(g, pre + 'pol:')
        assert sorted(pol) == sorted(want)
        for k, v in pol.items():
            assert np.allclose(v, want[k], atol=1e-6), k
        for k, v in vf.items():
            assert np.allclose(v, g[pre + 'vf:' + k], atol=1e-6), k
        for which, name in (('policy', 'pol'), ('vf', 'vf')):
            for j, (step, m, v) in enumerate(algo.adam_state(which)):
                assert step == int(g['%sadam_%s_%d_step' % (pre, name, j)])
                assert np.allclose(m, cg.flat(
                    g['%sadam_%s_%d_m' % (pre, name, j)]), atol=1e-7)
                assert np.allclose(v, cg.flat(
                    g['%sadam_%s_%d_v' % (pre, name, j)]), atol=1e-9)


def test_golden_normalized_env(golden):
    g = golden('normalized_env')
    norm = osamp.NormalizedObs(3, float(g['alpha']))
    for raw, normed, mean, var in zip(g['raw'], g['normed'], g['means'],
                                      g['variances']):
        got = norm(raw)
        assert np.allclose(got, normed, rtol=1e-12)
        assert np.allclose(norm.mean, mean, rtol=1e-12)
        assert np.allclose(norm.var, var, rtol=1e-12)


def test_golden_log_performance(golden):
    g = golden('log_performance')
    for tag in ('mixed', 'timeout'):
        lens = g[tag + '_lengths']
        S = int(lens.sum())
        b = ob.OracleEpisodeBatch(observations=np.zeros((S, 3)),
                                  last_observations=np.zeros((len(lens), 3)),
                                  actions=np.zeros((S, 2)),
                                  rewards=g[tag + '_rewards'],
                                  step_types=g[tag + '_step_types'],
                                  lengths=lens)
        stats, und = ob.performance_stats(b, 0.9)
        assert np.allclose(und, g[tag + '_undiscounted'])
        for k, v in stats.items():
            assert np.isclose(v, float(g[tag + ':Evaluation/' + k])), k


MULTITASK_CASES = {
    'named': (True, None),
    'ids': (False, {0: 'zero', 1: 'one', 5: 'five'}),
    'ids_nomap': (False, None),
}


def multitask_envs(g, use_names):
    """The environments tests/golden/make_golden.py::gen_multitask stepped."""
    P, n = [int(v) for v in g['cfg']]
    cyc = g['cycles']
    names = ['reach', 'push', 'reach', 'pick']
    return [
        oenvs.TaskEnv(i, cyc[i], P, task_id=i % 2,
                      task_name=names[i] if use_names else None,
                      success_at=[2, None, 5, 1][i]) for i in range(n)
    ], P, n


def check_multitask(g, tag, recorded, undiscounted):
    keys = [str(k) for k in g[tag + '_keys']]
    assert list(recorded) == keys  # same rows, same order
    for k, want in zip(keys, g[tag + '_vals']):
        got = float(recorded[k])
        assert (np.isnan(got) and np.isnan(want)) or np.isclose(got, want), k
    assert np.allclose(undiscounted, g[tag + '_undiscounted'])


@pytest.mark.parametrize('tag', sorted(MULTITASK_CASES))
def test_golden_multitask_env_infos_and_performance(golden, tag):
    """env_infos through the VecWorker restatement and the rows
    ``log_multitask_performance`` records, against the real reference
    (including its dropped per-task rows for task ids without a name map)."""
    g = golden('multitask')
    use_names, name_map = MULTITASK_CASES[tag]
    envs, P, n = multitask_envs(g, use_names)
    s = osamp.OracleLocalSampler(_Scripted(2), [envs], max_episode_length=P,
                                 n_workers=1,
                                 worker_class=osamp.OracleVecWorker,
                                 worker_args=dict(n_envs=n, alias_bug=True))
    eps = s.obtain_samples(0, 40, None)
    _check_batch(g, tag + '_', eps)
    want_keys = {k[len(tag) + 5:] for k in g.files
                 if k.startswith(tag + '_env_')}
    assert set(eps.env_infos) == want_keys
    for k in want_keys:
        assert np.array_equal(eps.env_infos[k], g[tag + '_env_' + k]), k
    rec, und = ob.multitask_performance_stats(7, eps, 0.9, name_map=name_map)
    check_multitask(g, tag, rec, und)


TRPO_CASES = {
    'trpo': {},
    'trpo_tight': {},
    'trpo_reg': dict(entropy_method='regularized', policy_ent_coeff=0.02),
    'trpo_reject': {},
}


@pytest.mark.parametrize('tag', sorted(TRPO_CASES))
def test_golden_trpo_train_once(golden, tag):
    """Two real TRPO ``_train_once`` iterations: CG
    direction, descent step, accepted step and post-step parameters."""
    from oracle.trpo import OracleTRPO
    g = golden('trpo_train_once')
    O, A, P, E, mb = [int(v) for v in g[tag + '_cfg']]
    algo = OracleTRPO(_params(g, tag + '_pol0:'), _params(g, tag + '_vf0:'),
                      max_episode_length=P,
                      max_constraint_value=float(g[tag + '_delta']),
                      max_backtracks=int(g[tag + '_max_backtracks']),
                      max_optimization_epochs=E, minibatch_size=mb,
                      **TRPO_CASES[tag])
    for it in range(2):
        pre = '%s_it%d_' % (tag, it)
        lens = g[pre + 'lengths']
        b = ob.OracleEpisodeBatch(
            observations=g[pre + 'observations'],
            last_observations=np.zeros((len(lens), O), np.float32),
            actions=g[pre + 'actions'], rewards=g[pre + 'rewards'],
            step_types=g[pre + 'step_types'], lengths=lens,
            max_episode_length=P)
        np.random.seed(int(g[pre + 'np_seed']))
        out = algo.train_once(b)
        tr = algo.cg.trace
        # (the second iteration starts from parameters that already carry the
        # first step's CG rounding, ~1e-5)
        assert np.allclose(tr['grad'], g[pre + 'cg:grad'],
                           atol=1e-6 if it == 0 else 5e-5)
        scale = np.abs(g[pre + 'cg:step_dir']).max()
        assert np.allclose(tr['step_dir'], g[pre + 'cg:step_dir'],
                           atol=(1e-4 if it == 0 else 2e-3) * scale)
        # ten fp32 CG iterations amplify last-ulp differences of the gradient
        # to ~1e-4 of the direction: the tolerance the reference itself has
        dscale = np.abs(g[pre + 'cg:descent_step']).max()
        assert np.allclose(tr['descent_step'], g[pre + 'cg:descent_step'],
                           atol=(5e-4 if it == 0 else 2e-3) * dscale)
        for mine, theirs in LOG_KEYS.items():
            assert np.isclose(out[mine], float(g[pre + 'log:' + theirs]),
                              atol=1e-5 if it == 0 else 1e-4,
                              rtol=1e-4), (mine, it)
        pol, vf = algo.state()
        for k, v in pol.items():
            assert np.allclose(v, g[pre + 'pol:' + k],
                               atol=(5e-4 if it == 0 else 2e-3) * dscale), k
        for k, v in vf.items():
            assert np.allclose(v, g[pre + 'vf:' + k], atol=1e-6), k
    if tag == 'trpo_reject':
        assert algo.cg.trace['accepted'] in (-1, 0)


@pytest.mark.parametrize('tag', sorted(TRPO_CASES))
def test_golden_trpo_per_iterate_pins(golden, tag):
    """The end-to-end comparison above is as loose as ten fp32 CG iterations make
    it (the reference run is itself one sample of that rounding noise).  Each
    operation of the step is pinned on its own instead, at the reference's OWN
    iterates: the Hessian-vector product ``A p_k`` for every direction ``p_k`` the
    real ``_conjugate_gradient`` visited
    (``conjugate_gradient_optimizer.py:69-104``), and the (loss, constraint) pair
    of every backtracking candidate of the real descent step (``:236-277``) -- all
    at one-operation fp32 accuracy."""
    from oracle.trpo import OracleTRPO
    g = golden('trpo_train_once')
    O, A, P, E, mb = [int(v) for v in g[tag + '_cfg']]
    algo = OracleTRPO(_params(g, tag + '_pol0:'), _params(g, tag + '_vf0:'),
                      max_episode_length=P,
                      max_constraint_value=float(g[tag + '_delta']),
                      max_backtracks=int(g[tag + '_max_backtracks']),
                      max_optimization_epochs=E, minibatch_size=mb,
                      **TRPO_CASES[tag])
    pre = tag + '_it0_'
    algo.cg.probe_vectors = g[pre + 'cg:iter_p']
    algo.cg.probe_descent = g[pre + 'cg:descent_step']
    algo.cg.probe_candidates = len(g[pre + 'cg:ls_constraint'])
    lens = g[pre + 'lengths']
    b = ob.OracleEpisodeBatch(
        observations=g[pre + 'observations'],
        last_observations=np.zeros((len(lens), O), np.float32),
        actions=g[pre + 'actions'], rewards=g[pre + 'rewards'],
        step_types=g[pre + 'step_types'], lengths=lens, max_episode_length=P)
    np.random.seed(int(g[pre + 'np_seed']))
    algo.train_once(b)
    tr = algo.cg.trace
    want = g[pre + 'cg:iter_Ap']
    assert tr['probe_Ax'].shape == want.shape and len(want) == 10
    for k in range(len(want)):
        scale = np.abs(want[k]).max()
        assert np.allclose(tr['probe_Ax'][k], want[k], atol=1e-5 * scale,
                           rtol=1e-5), k
    ls = tr['probe_ls']
    assert np.isclose(ls[0], g[pre + 'cg:ls_loss'][0], atol=1e-7)
    for k, (loss, kl) in enumerate(ls[1:]):
        assert np.isclose(loss, g[pre + 'cg:ls_loss'][k + 1], atol=1e-6,
                          rtol=1e-5), k
        assert np.isclose(kl, g[pre + 'cg:ls_constraint'][k], atol=1e-7,
                          rtol=1e-4), k


TRPO_CATEGORICAL_CASES = {
    'trpo': {},
    'trpo3': {},
    'trpo_reg': dict(entropy_method='regularized', policy_ent_coeff=0.02),
    'trpo_c2': {},
}


def _categorical_trpo(g, tag):
    import _categorical_golden as cg
    from oracle.trpo import OracleTRPO
    O, n_act, P, E, mb = [int(v) for v in g[tag + '_cfg']]
    algo = OracleTRPO(cg.policy_params(g, tag + '_pol0:'),
                      cg.value_params(g, tag + '_vf0:'),
                      max_episode_length=P, policy_kind='categorical',
                      max_constraint_value=float(g[tag + '_delta']),
                      max_backtracks=int(g[tag + '_max_backtracks']),
                      max_optimization_epochs=E, minibatch_size=mb,
                      **TRPO_CATEGORICAL_CASES[tag])
    return algo, O, P


def _categorical_trpo_batch(g, pre, O, P):
    lens = g[pre + 'lengths']
    return ob.OracleEpisodeBatch(
        observations=g[pre + 'observations'],
        last_observations=np.zeros((len(lens), O), np.float32),
        actions=g[pre + 'actions'], rewards=g[pre + 'rewards'],
        step_types=g[pre + 'step_types'], lengths=lens, max_episode_length=P)


@pytest.mark.parametrize('tag', sorted(TRPO_CATEGORICAL_CASES))
def test_golden_trpo_categorical_train_once(golden, tag):
    """Two real TRPO iterations on the reference's ``CategoricalCNNPolicy``
    configured as an MLP (tests/_categorical_golden.py; ``gen_trpo_categorical``).

    End to end this comparison is LOOSE by nature, looser than the Gaussian one:
    class probabilities do not change when every score moves by the same amount,
    so the KL Hessian of a softmax head has a null direction (the output biases
    along (1, ..., 1)) that only ``hvp_reg_coeff`` = 1e-5 keeps finite, and ten
    fp32 conjugate-gradient iterations amplify last-bit differences (here: the
    reference's first layer is a 1 x 1 convolution, the oracle's a linear layer)
    to 0.05-2 % of the direction.  The reference's run is one sample of that
    noise.  What is tight is the gradient the iteration starts from, the number
    of backtracking candidates, and ``test_golden_trpo_categorical_per_iterate_
    pins`` below: every operation at the reference's own iterates, 1e-5."""
    import _categorical_golden as cg
    g = golden('trpo_categorical')
    algo, O, P = _categorical_trpo(g, tag)
    for it in range(2):
        pre = '%s_it%d_' % (tag, it)
        np.random.seed(int(g[pre + 'np_seed']))
        out = algo.train_once(_categorical_trpo_batch(g, pre, O, P))
        tr = algo.cg.trace
        if it == 0:
            assert np.allclose(tr['grad'], g[pre + 'cg:grad'], atol=1e-6)
            assert tr['accepted'] + 1 == len(g[pre + 'cg:ls_constraint'])
        loose = 3e-2 if it == 0 else 6e-2
        scale = np.abs(g[pre + 'cg:step_dir']).max()
        assert np.allclose(tr['step_dir'], g[pre + 'cg:step_dir'],
                           atol=loose * scale)
        dscale = np.abs(g[pre + 'cg:descent_step']).max()
        assert np.allclose(tr['descent_step'], g[pre + 'cg:descent_step'],
                           atol=loose * dscale)
        for mine, theirs in cg.LOG_KEYS.items():
            assert np.isclose(out[mine], float(g[pre + 'log:' + theirs]),
                              atol=2e-4 if it == 0 else 1e-3,
                              rtol=5e-2), (mine, it)
        pol, vf = algo.state()
        want = cg.policy_params(g, pre + 'pol:')
        for k, v in pol.items():
            assert np.allclose(v, want[k], atol=loose * dscale), k
        for k, v in vf.items():
            assert np.allclose(v, g[pre + 'vf:' + k], atol=1e-6), k


@pytest.mark.parametrize('tag', sorted(TRPO_CATEGORICAL_CASES))
def test_golden_trpo_categorical_per_iterate_pins(golden, tag):
    """Every Hessian-vector product and line-search candidate of the real run,
    each at one-operation accuracy (see ``test_golden_trpo_per_iterate_pins``)."""
    g = golden('trpo_categorical')
    algo, O, P = _categorical_trpo(g, tag)
    pre = tag + '_it0_'
    algo.cg.probe_vectors = g[pre + 'cg:iter_p']
    algo.cg.probe_descent = g[pre + 'cg:descent_step']
    algo.cg.probe_candidates = len(g[pre + 'cg:ls_constraint'])
    np.random.seed(int(g[pre + 'np_seed']))
    algo.train_once(_categorical_trpo_batch(g, pre, O, P))
    tr = algo.cg.trace
    want = g[pre + 'cg:iter_Ap']
    assert tr['probe_Ax'].shape == want.shape and len(want) == 10
    for k in range(len(want)):
        scale = np.abs(want[k]).max()
        assert np.allclose(tr['probe_Ax'][k], want[k], atol=1e-5 * scale,
                           rtol=1e-5), k
    ls = tr['probe_ls']
    assert np.isclose(ls[0], g[pre + 'cg:ls_loss'][0], atol=1e-7)
    for k, (loss, kl) in enumerate(ls[1:]):
        assert np.isclose(loss, g[pre + 'cg:ls_loss'][k + 1], atol=1e-6,
                          rtol=1e-5), k
        assert np.isclose(kl, g[pre + 'cg:ls_constraint'][k], atol=1e-7,
                          rtol=1e-4), k


POLICY_OPTION_CASES = ['fixed_std', 'init_small', 'max_clamp', 'min_clamp']


@pytest.mark.parametrize('tag', POLICY_OPTION_CASES)
def test_golden_policy_std_options(golden, tag):
    """Two real PPO iterations per ``GaussianMLPPolicy`` std option: fixed std
    (a buffer, not a parameter), active max / min clamps (zero gradient through
    the clamp), a small learned std."""
    g = golden('policy_options')
    O, A, P, E, mb = [int(v) for v in g[tag + '_cfg']]
    algo = OraclePPO(_params(g, tag + '_pol0:'), _params(g, tag + '_vf0:'),
                     max_episode_length=P, max_optimization_epochs=E,
                     minibatch_size=mb, policy_lr=2.5e-3, vf_lr=2.5e-3)
    for it in range(2):
        pre = '%s_it%d_' % (tag, it)
        lens = g[pre + 'lengths']
        b = ob.OracleEpisodeBatch(
            observations=g[pre + 'observations'],
            last_observations=np.zeros((len(lens), O), np.float32),
            actions=g[pre + 'actions'], rewards=g[pre + 'rewards'],
            step_types=g[pre + 'step_types'], lengths=lens,
            max_episode_length=P)
        np.random.seed(int(g[pre + 'np_seed']))
        out = algo.train_once(b)
        for mine, theirs in LOG_KEYS.items():
            assert np.isclose(out[mine], float(g[pre + 'log:' + theirs]),
                              atol=1e-5, rtol=1e-5), (mine, it)
        pol, vf = algo.state()
        for k, v in pol.items():
            assert np.allclose(v, g[pre + 'pol:' + k], atol=1e-6), k
        for k, v in vf.items():
            assert np.allclose(v, g[pre + 'vf:' + k], atol=1e-6), k


ACTIVATION_CASES = {
    'relu': (torch.relu, torch.relu),
    'linear': (None, None),
    'relu_policy_tanh_vf': (torch.relu, torch.tanh),
}


@pytest.mark.parametrize('tag', sorted(ACTIVATION_CASES))
def test_golden_hidden_nonlinearities(golden, tag):
    """``hidden_nonlinearity`` = relu / None (``mlp_module.py:43-44``): forward
    outputs of freshly built real networks and two real PPO iterations."""
    from oracle import networks as nets
    g = golden('policy_options')
    O, A, P, E, mb = [int(v) for v in g[tag + '_cfg']]
    pa, va = ACTIVATION_CASES[tag]
    with nets.hidden_nonlinearity(policy=pa, value=va):
        pol0, vf0 = _params(g, tag + '_pol0:'), _params(g, tag + '_vf0:')
        x = torch.from_numpy(g[tag + '_fwd_obs'])
        with torch.no_grad():
            assert np.allclose(nets.policy_forward(pol0, x)[0].mean.numpy(),
                               g[tag + '_fwd_mean'], atol=1e-6)
            assert np.allclose(nets.value_forward(vf0, x).numpy(),
                               g[tag + '_fwd_value'], atol=1e-6)
        algo = OraclePPO(pol0, vf0, max_episode_length=P,
                         max_optimization_epochs=E, minibatch_size=mb,
                         policy_lr=2.5e-3, vf_lr=2.5e-3)
        for it in range(2):
            pre = '%s_it%d_' % (tag, it)
            lens = g[pre + 'lengths']
            b = ob.OracleEpisodeBatch(
                observations=g[pre + 'observations'],
                last_observations=np.zeros((len(lens), O), np.float32),
                actions=g[pre + 'actions'], rewards=g[pre + 'rewards'],
                step_types=g[pre + 'step_types'], lengths=lens,
                max_episode_length=P)
            np.random.seed(int(g[pre + 'np_seed']))
            out = algo.train_once(b)
            for mine, theirs in LOG_KEYS.items():
                assert np.isclose(out[mine], float(g[pre + 'log:' + theirs]),
                                  atol=1e-5, rtol=1e-5), (mine, it)
            pol, vf = algo.state()
            for k, v in pol.items():
                assert np.allclose(v, g[pre + 'pol:' + k], atol=1e-6), k
            for k, v in vf.items():
                assert np.allclose(v, g[pre + 'vf:' + k], atol=1e-6), k


@pytest.mark.parametrize('tag', ['softplus', 'softplus_max_clamp'])
def test_golden_softplus_std(golden, tag):
    """``std_parameterization='softplus'``: std = log(1 + exp(exp(p)))
    (``gaussian_mlp_module.py:178-181``), free and with an active upper clamp on
    ``p`` (zero gradient): forward log-std and two real PPO iterations."""
    from oracle import networks as nets
    g = golden('policy_options')
    O, A, P, E, mb = [int(v) for v in g[tag + '_cfg']]
    with nets.std_parameterization('softplus'):
        pol0, vf0 = _params(g, tag + '_pol0:'), _params(g, tag + '_vf0:')
        x = torch.from_numpy(g[tag + '_fwd_obs'])
        with torch.no_grad():
            dist, info = nets.policy_forward(pol0, x)
            assert np.allclose(dist.mean.numpy(), g[tag + '_fwd_mean'], atol=1e-6)
            assert np.allclose(info['log_std'].numpy(), g[tag + '_fwd_log_std'],
                               atol=1e-6)
        algo = OraclePPO(pol0, vf0, max_episode_length=P,
                         max_optimization_epochs=E, minibatch_size=mb,
                         policy_lr=2.5e-3, vf_lr=2.5e-3)
        for it in range(2):
            pre = '%s_it%d_' % (tag, it)
            lens = g[pre + 'lengths']
            b = ob.OracleEpisodeBatch(
                observations=g[pre + 'observations'],
                last_observations=np.zeros((len(lens), O), np.float32),
                actions=g[pre + 'actions'], rewards=g[pre + 'rewards'],
                step_types=g[pre + 'step_types'], lengths=lens,
                max_episode_length=P)
            np.random.seed(int(g[pre + 'np_seed']))
            out = algo.train_once(b)
            for mine, theirs in LOG_KEYS.items():
                assert np.isclose(out[mine], float(g[pre + 'log:' + theirs]),
                                  atol=1e-5, rtol=1e-5), (mine, it)
            pol, vf = algo.state()
            for k, v in pol.items():
                assert np.allclose(v, g[pre + 'pol:' + k], atol=1e-6), k
            for k, v in vf.items():
                assert np.allclose(v, g[pre + 'vf:' + k], atol=1e-6), k


OUTPUT_CASES = {
    # tag: (policy hidden, policy output, value hidden, value output)
    'out_tanh': (torch.tanh, torch.tanh, torch.tanh, None),
    'out_tanh_vf_relu_hidden': (torch.relu, torch.tanh, torch.tanh, torch.tanh),
}


@pytest.mark.parametrize('tag', sorted(OUTPUT_CASES))
def test_golden_output_nonlinearity(golden, tag):
    """``output_nonlinearity`` on the Gaussian mean / the value
    (``mlp_module.py:52-53``): forward outputs and two real PPO iterations."""
    from oracle import networks as nets
    g = golden('policy_options')
    O, A, P, E, mb = [int(v) for v in g[tag + '_cfg']]
    ph, po, vh, vo = OUTPUT_CASES[tag]
    with nets.hidden_nonlinearity(policy=ph, value=vh), \
            nets.output_nonlinearity(policy=po, value=vo):
        pol0, vf0 = _params(g, tag + '_pol0:'), _params(g, tag + '_vf0:')
        x = torch.from_numpy(g[tag + '_fwd_obs'])
        with torch.no_grad():
            assert np.allclose(nets.policy_forward(pol0, x)[0].mean.numpy(),
                               g[tag + '_fwd_mean'], atol=1e-6)
            assert np.allclose(nets.value_forward(vf0, x).numpy(),
                               g[tag + '_fwd_value'], atol=1e-6)
        algo = OraclePPO(pol0, vf0, max_episode_length=P,
                         max_optimization_epochs=E, minibatch_size=mb,
                         policy_lr=2.5e-3, vf_lr=2.5e-3)
        for it in range(2):
            pre = '%s_it%d_' % (tag, it)
            lens = g[pre + 'lengths']
            b = ob.OracleEpisodeBatch(
                observations=g[pre + 'observations'],
                last_observations=np.zeros((len(lens), O), np.float32),
                actions=g[pre + 'actions'], rewards=g[pre + 'rewards'],
                step_types=g[pre + 'step_types'], lengths=lens,
                max_episode_length=P)
            np.random.seed(int(g[pre + 'np_seed']))
            out = algo.train_once(b)
            for mine, theirs in LOG_KEYS.items():
                assert np.isclose(out[mine], float(g[pre + 'log:' + theirs]),
                                  atol=1e-5, rtol=1e-5), (mine, it)
            pol, vf = algo.state()
            for k, v in pol.items():
                assert np.allclose(v, g[pre + 'pol:' + k], atol=1e-6), k
            for k, v in vf.items():
                assert np.allclose(v, g[pre + 'vf:' + k], atol=1e-6), k


# round 3: more callables of the reference's NonLinearity wrapper
# (tests/golden/policy_activations.npz): tag -> (policy hidden, policy output,
# value hidden, value output)
MORE_ACTIVATION_CASES = {
    'sigmoid': (torch.sigmoid, None, torch.sigmoid, None),
    'elu': (torch.nn.functional.elu, None, torch.nn.functional.elu, None),
    'leaky_relu': (torch.nn.functional.leaky_relu, None,
                   torch.nn.functional.leaky_relu, None),
    'softplus_hidden': (torch.nn.functional.softplus, None,
                        torch.nn.functional.softplus, None),
    'out_sigmoid_elu_hidden': (torch.nn.functional.elu, torch.sigmoid,
                               torch.tanh, torch.nn.functional.softplus),
}


@pytest.mark.parametrize('tag', sorted(MORE_ACTIVATION_CASES))
def test_golden_more_activations(golden, tag):
    """sigmoid / elu / leaky_relu / softplus as ``hidden_nonlinearity`` or
    ``output_nonlinearity`` (``multi_headed_mlp_module.py:154-197``): forward
    outputs of freshly built real networks and two real PPO iterations."""
    from oracle import networks as nets
    g = golden('policy_activations')
    O, A, P, E, mb = [int(v) for v in g[tag + '_cfg']]
    ph, po, vh, vo = MORE_ACTIVATION_CASES[tag]
    with nets.hidden_nonlinearity(policy=ph, value=vh), \
            nets.output_nonlinearity(policy=po, value=vo):
        pol0, vf0 = _params(g, tag + '_pol0:'), _params(g, tag + '_vf0:')
        x = torch.from_numpy(g[tag + '_fwd_obs'])
        with torch.no_grad():
            assert np.allclose(nets.policy_forward(pol0, x)[0].mean.numpy(),
                               g[tag + '_fwd_mean'], atol=1e-6)
            assert np.allclose(nets.value_forward(vf0, x).numpy(),
                               g[tag + '_fwd_value'], atol=1e-6)
        algo = OraclePPO(pol0, vf0, max_episode_length=P,
                         max_optimization_epochs=E, minibatch_size=mb,
                         policy_lr=2.5e-3, vf_lr=2.5e-3)
        for it in range(2):
            pre = '%s_it%d_' % (tag, it)
            lens = g[pre + 'lengths']
            b = ob.OracleEpisodeBatch(
                observations=g[pre + 'observations'],
                last_observations=np.zeros((len(lens), O), np.float32),
                actions=g[pre + 'actions'], rewards=g[pre + 'rewards'],
                step_types=g[pre + 'step_types'], lengths=lens,
                max_episode_length=P)
            np.random.seed(int(g[pre + 'np_seed']))
            out = algo.train_once(b)
            for mine, theirs in LOG_KEYS.items():
                assert np.isclose(out[mine], float(g[pre + 'log:' + theirs]),
                                  atol=1e-5, rtol=1e-5), (mine, it)
            pol, vf = algo.state()
            for k, v in pol.items():
                assert np.allclose(v, g[pre + 'pol:' + k], atol=1e-6), k
            for k, v in vf.items():
                assert np.allclose(v, g[pre + 'vf:' + k], atol=1e-6), k


# round 3: torch.optim classes other than the default Adam through make_optimizer
# (tests/golden/train_once_optimizers.npz)
OPTIMIZER_CASES = {
    'sgd_plain': (torch.optim.SGD, dict(lr=5e-2)),
    'sgd_nesterov_wd': (torch.optim.SGD, dict(lr=2e-2, momentum=0.9,
                                              nesterov=True, weight_decay=1e-3)),
    'sgd_momentum_dampening': (torch.optim.SGD, dict(lr=2e-2, momentum=0.8,
                                                     dampening=0.1)),
    'rmsprop': (torch.optim.RMSprop, dict(lr=1e-3)),
    'rmsprop_centered_momentum': (torch.optim.RMSprop,
                                  dict(lr=1e-3, alpha=0.9, momentum=0.5,
                                       centered=True, weight_decay=1e-3)),
    'adam_amsgrad_wd': (torch.optim.Adam, dict(lr=2.5e-3, amsgrad=True,
                                               weight_decay=1e-2)),
    'adamw': (torch.optim.AdamW, dict(lr=2.5e-3, weight_decay=5e-2)),
}


@pytest.mark.parametrize('tag', sorted(OPTIMIZER_CASES))
def test_golden_train_once_other_optimizers(golden, tag):
    """``OptimizerWrapper((torch.optim.X, kwargs), module)`` for X other than the
    default Adam (``make_optimizer``, ``_functions.py:25-65``): two real PPO
    iterations, logged scalars and post-update parameters."""
    g = golden('train_once_optimizers')
    O, A, P, E, mb = [int(v) for v in g[tag + '_cfg']]
    cls, kw = OPTIMIZER_CASES[tag]
    pol0, vf0 = _params(g, tag + '_pol0:'), _params(g, tag + '_vf0:')
    algo = OraclePPO(pol0, vf0, max_episode_length=P, max_optimization_epochs=E,
                     minibatch_size=mb, policy_optimizer=(cls, dict(kw)),
                     vf_optimizer=(cls, dict(kw)))
    for it in range(2):
        pre = '%s_it%d_' % (tag, it)
        lens = g[pre + 'lengths']
        b = ob.OracleEpisodeBatch(
            observations=g[pre + 'observations'],
            last_observations=np.zeros((len(lens), O), np.float32),
            actions=g[pre + 'actions'], rewards=g[pre + 'rewards'],
            step_types=g[pre + 'step_types'], lengths=lens,
            max_episode_length=P)
        np.random.seed(int(g[pre + 'np_seed']))
        out = algo.train_once(b)
        for mine, theirs in LOG_KEYS.items():
            assert np.isclose(out[mine], float(g[pre + 'log:' + theirs]),
                              atol=1e-5, rtol=1e-5), (mine, it)
        pol, vf = algo.state()
        for k, v in pol.items():
            assert np.allclose(v, g[pre + 'pol:' + k], atol=1e-6), k
        for k, v in vf.items():
            assert np.allclose(v, g[pre + 'vf:' + k], atol=1e-6), k


LAYER_NORM_CASES = {
    'layer_norm': (torch.tanh, torch.tanh),
    'layer_norm_relu': (torch.relu, torch.tanh),
}


@pytest.mark.parametrize('tag', sorted(LAYER_NORM_CASES))
def test_golden_layer_normalization(golden, tag):
    """``layer_normalization=True`` (``multi_headed_mlp_module.py:77-81``: a
    LayerNorm over the input of every hidden linear layer): forward outputs and two
    real PPO iterations, gamma / beta included."""
    from oracle import networks as nets
    g = golden('policy_options')
    O, A, P, E, mb = [int(v) for v in g[tag + '_cfg']]
    pa, va = LAYER_NORM_CASES[tag]
    with nets.hidden_nonlinearity(policy=pa, value=va):
        pol0, vf0 = _params(g, tag + '_pol0:'), _params(g, tag + '_vf0:')
        assert any('layer_normalization' in k for k in pol0)
        x = torch.from_numpy(g[tag + '_fwd_obs'])
        with torch.no_grad():
            assert np.allclose(nets.policy_forward(pol0, x)[0].mean.numpy(),
                               g[tag + '_fwd_mean'], atol=1e-6)
            assert np.allclose(nets.value_forward(vf0, x).numpy(),
                               g[tag + '_fwd_value'], atol=1e-6)
        algo = OraclePPO(pol0, vf0, max_episode_length=P,
                         max_optimization_epochs=E, minibatch_size=mb,
                         policy_lr=2.5e-3, vf_lr=2.5e-3)
        for it in range(2):
            pre = '%s_it%d_' % (tag, it)
            lens = g[pre + 'lengths']
            b = ob.OracleEpisodeBatch(
                observations=g[pre + 'observations'],
                last_observations=np.zeros((len(lens), O), np.float32),
                actions=g[pre + 'actions'], rewards=g[pre + 'rewards'],
                step_types=g[pre + 'step_types'], lengths=lens,
                max_episode_length=P)
            np.random.seed(int(g[pre + 'np_seed']))
            out = algo.train_once(b)
            for mine, theirs in LOG_KEYS.items():
                assert np.isclose(out[mine], float(g[pre + 'log:' + theirs]),
                                  atol=1e-5, rtol=1e-5), (mine, it)
            pol, vf = algo.state()
            for k, v in pol.items():
                assert np.allclose(v, g[pre + 'pol:' + k], atol=1e-6), k
            for k, v in vf.items():
                assert np.allclose(v, g[pre + 'vf:' + k], atol=1e-6), k
