"""End-to-end parity on the GPU: sampler bookkeeping and PPO/VPG iterations
against goldens captured from the real reference and against the oracle."""
from collections import OrderedDict

import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

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

TRAIN_CASES = {
    'ppo': dict(),
    'ppo_pos': dict(positive_adv=True),
    'ppo_reg': dict(entropy_method='regularized', policy_ent_coeff=0.02),
    'ppo_max': dict(entropy_method='max', policy_ent_coeff=0.05,
                    center_adv=False, stop_entropy_gradient=True,
                    use_softplus_entropy=True),
    'vpg': dict(),
    'ppo_full': dict(),
}


def _sd(g, prefix):
    out = OrderedDict()
    for k in g.files:
        if k.startswith(prefix):
            out[k[len(prefix):]] = torch.from_numpy(g[k].copy())
    return out


def _spec(O, A, P):
    from garage_amd._dtypes import Box, EnvSpec
    return EnvSpec(Box(-np.inf, np.inf, (O, )), Box(-np.inf, np.inf, (A, )),
                   max_episode_length=P)


def _host_batch(spec, g, pre, O):
    from garage_amd._dtypes import EpisodeBatch, StepType
    lens = g[pre + 'lengths']
    st = np.asarray([StepType(int(s)) for s in g[pre + 'step_types']],
                    dtype=object)
    return EpisodeBatch(env_spec=spec, episode_infos={},
                        observations=g[pre + 'observations'],
                        last_observations=np.zeros((len(lens), O), np.float32),
                        actions=g[pre + 'actions'], rewards=g[pre + 'rewards'],
                        env_infos={}, agent_infos={}, step_types=st,
                        lengths=lens)


@pytest.mark.parametrize('tag', sorted(TRAIN_CASES))
def test_train_once_matches_real_reference(golden, tag):
    """Two consecutive ``_train_once`` calls vs the real garage PPO / VPG
    (tests/golden/train_once.npz): 9 logged scalars, evaluation statistics,
    post-update parameters and Adam moments."""
    from garage_amd.algos import PPO, VPG
    from garage_amd.optimizers import OptimizerWrapper
    from garage_amd.policies import GaussianMLPPolicy, GaussianMLPValueFunction
    g = golden('train_once')
    O, A, P, E, mb = [int(v) for v in g[tag + '_cfg']]
    mb = None if mb < 0 else mb
    spec = _spec(O, A, P)
    pol = GaussianMLPPolicy(spec, hidden_sizes=(8, 8))
    vf = GaussianMLPValueFunction(spec, hidden_sizes=(8, 8))
    pol.load_state_dict(_sd(g, tag + '_pol0:'))
    vf.load_state_dict(_sd(g, tag + '_vf0:'))
    cls = VPG if tag == 'vpg' else PPO
    algo = cls(env_spec=spec, policy=pol, value_function=vf, sampler=None,
               policy_optimizer=OptimizerWrapper(
                   (torch.optim.Adam, dict(lr=2.5e-4)), pol,
                   max_optimization_epochs=E, minibatch_size=mb),
               vf_optimizer=OptimizerWrapper(
                   (torch.optim.Adam, dict(lr=2.5e-4)), vf,
                   max_optimization_epochs=E, minibatch_size=mb),
               **TRAIN_CASES[tag])
    for it in range(2):
        pre = '%s_it%d_' % (tag, it)
        batch = _host_batch(spec, g, pre, O)
        np.random.seed(int(g[pre + 'np_seed']))
        avg = algo._train_once(it, batch)
        for mine, theirs in LOG_KEYS.items():
            want = float(g[pre + 'log:' + theirs])
            assert np.isclose(algo.last_tabular[mine], want, atol=1e-5,
                              rtol=1e-5), (mine, it, algo.last_tabular[mine],
                                           want)
        assert np.isclose(avg, float(g[pre + 'avg_return']))
        for k, v in algo.last_performance.items():
            assert np.isclose(v, float(g[pre + 'log:Evaluation/' + k])), k
        for k, v in pol.state_dict().items():
            assert np.allclose(v.numpy(), g[pre + 'pol:' + k], atol=2e-6), k
        for k, v in vf.state_dict().items():
            assert np.allclose(v.numpy(), g[pre + 'vf:' + k], atol=2e-6), k
        for name, net in (('pol', pol.net), ('vf', vf.net)):
            views_m = net.named_views(net.exp_avg)
            views_v = net.named_views(net.exp_avg_sq)
            for j, ((_, m), (_, v)) in enumerate(zip(views_m, views_v)):
                gm = g['%sadam_%s_%d_m' % (pre, name, j)]
                gv = g['%sadam_%s_%d_v' % (pre, name, j)]
                assert np.allclose(m.cpu().numpy().reshape(gm.shape), gm,
                                   atol=1e-6)
                assert np.allclose(v.cpu().numpy().reshape(gv.shape), gv,
                                   atol=1e-8)
            assert net.adam_steps == int(g['%sadam_%s_0_step' % (pre, name)])


def _scripted_policy(spec):
    """action = [sum(obs), step]: a linear 'MLP' + teacher-forced noise."""
    from garage_amd.policies import GaussianMLPPolicy
    pol = GaussianMLPPolicy(spec, hidden_sizes=(), init_std=1.0)
    pol.net.weight(0).copy_(torch.tensor([[1., 1., 1.], [0., 0., 0.]]))
    pol.net.bias(0).zero_()
    return pol


def test_sampler_bookkeeping_matches_real_vecworker(golden):
    """lengths / order / step types / rewards / actions / last_observations
    equal the real ``LocalSampler(VecWorker)`` (tests/golden/sampler.npz);
    observations equal what ``DefaultWorker`` semantics give (Q10)."""
    from garage_amd.sampler import GpuVecSampler, GpuVecWorker
    from oracle import envs as oenvs
    from oracle import sampler as osamp
    g = golden('sampler')
    P, n = [int(v) for v in g['cfg']]
    cyc = g['cycles']
    spec = _spec(3, 2, P)

    class Env(oenvs.CountingEnv):

        def __init__(self, i):
            super().__init__(i, cyc[i], P)
            self.spec = spec

    pol = _scripted_policy(spec)
    dev = pol.device

    def noise_fn(step):
        z = torch.zeros(n, 4, device=dev)
        z[:, 1] = float(step)
        return z

    sampler = GpuVecSampler(pol, [[Env(i) for i in range(n)]],
                            max_episode_length=P, n_workers=1,
                            worker_class=GpuVecWorker,
                            worker_args=dict(n_envs=n, noise_fn=noise_fn))
    # the oracle's VecWorker with DefaultWorker observations (alias bug off)
    class Scripted:
        calls = 0

        def reset(self, do_resets=None):
            pass

        def get_actions(self, obs):
            obs = np.asarray(obs, dtype=np.float32)
            a = np.zeros((obs.shape[0], 2), np.float32)
            a[:, 0] = obs.sum(axis=1)
            a[:, 1] = self.calls
            self.calls += 1
            return a, {}

    ref = osamp.OracleLocalSampler(
        Scripted(), [[oenvs.CountingEnv(i, cyc[i], P) for i in range(n)]],
        max_episode_length=P, n_workers=1, worker_class=osamp.OracleVecWorker,
        worker_args=dict(n_envs=n))
    for prefix, num in (('vec_', 30), ('vec2_', 17)):
        eps = sampler.obtain_samples(0, num, None)
        want = ref.obtain_samples(0, num, None)
        assert np.array_equal(eps.lengths, g[prefix + 'lengths'])
        assert eps.lengths.dtype == np.dtype('l')
        assert np.array_equal([int(s) for s in eps.step_types],
                              g[prefix + 'step_types'])
        assert np.array_equal(eps.rewards, g[prefix + 'rewards'])
        assert eps.rewards.dtype == np.float64
        assert np.array_equal(eps.actions, g[prefix + 'actions'])
        assert np.array_equal(eps.last_observations,
                              g[prefix + 'last_observations'])
        assert np.array_equal(eps.observations, want.observations)
        eps.to_host()  # passes the reference's EpisodeBatch validation
    assert sampler.total_env_steps == int(g['vec_total_env_steps'])


@pytest.mark.parametrize('ragged', [False, True])
def test_synthetic_rollout_matches_cpu_twin(ragged):
    """SyntheticVecEnv + GpuVecWorker vs the per-env CPU twin stepped by the
    oracle's VecWorker, teacher-forced with the same noise."""
    from garage_amd.envs import SyntheticVecEnv
    from garage_amd.policies import GaussianMLPPolicy
    from garage_amd.sampler import GpuVecSampler, GpuVecWorker
    from oracle import envs as oenvs
    from oracle import networks as nets
    from oracle import sampler as osamp
    n, O, A, P = 37, 5, 3, 12
    min_len = 3 if ragged else None
    torch.manual_seed(3)
    env = SyntheticVecEnv(n, O, A, P, min_len=min_len, seed=123)
    pol = GaussianMLPPolicy(env.spec, hidden_sizes=(16, 16))
    dev = pol.device
    noise = torch.randn(64, n, 4)

    def noise_fn(step):
        return noise[step].to(dev)

    sampler = GpuVecSampler(pol, env, max_episode_length=P, n_workers=1,
                            worker_class=GpuVecWorker,
                            worker_args=dict(n_envs=n, noise_fn=noise_fn))
    params = pol.state_dict()

    class CpuPolicy:
        calls = 0

        def reset(self, do_resets=None):
            pass

        def get_actions(self, obs):
            with torch.no_grad():
                dist, info = nets.policy_forward(
                    params, torch.from_numpy(np.asarray(obs, np.float32)))
            a = dist.mean + dist.stddev * noise[self.calls][:, :A]
            self.calls += 1
            return a.numpy(), {'mean': info['mean'].numpy()}

    ref = osamp.OracleLocalSampler(
        CpuPolicy(),
        [[oenvs.SyntheticEnv(i, O, A, P, min_len=min_len, seed=123)
          for i in range(n)]], max_episode_length=P, n_workers=1,
        worker_class=osamp.OracleVecWorker, worker_args=dict(n_envs=n))
    for num in (n * P, 150):
        eps = sampler.obtain_samples(0, num, None)
        want = ref.obtain_samples(0, num, None)
        assert np.array_equal(eps.lengths, want.lengths)
        assert np.array_equal([int(s) for s in eps.step_types],
                              [int(s) for s in want.step_types])
        assert np.array_equal(eps.observations, want.observations)  # bit exact
        assert np.array_equal(eps.last_observations, want.last_observations)
        assert np.allclose(eps.actions, want.actions, atol=1e-5)
        assert np.allclose(eps.rewards, want.rewards, atol=1e-5)
        assert np.allclose(eps.agent_infos['mean'], want.agent_infos['mean'],
                           atol=1e-5)
    assert sampler.total_env_steps == ref.total_env_steps


def test_full_iteration_vs_oracle():
    """GPU sampler -> PPO._train_once vs the oracle PPO on the same batch
    (ragged episodes, Adam over several minibatches, numpy permutations)."""
    from garage_amd.algos import PPO
    from garage_amd.envs import SyntheticVecEnv
    from garage_amd.optimizers import OptimizerWrapper
    from garage_amd.policies import GaussianMLPPolicy, GaussianMLPValueFunction
    from garage_amd.sampler import GpuVecSampler, GpuVecWorker
    from oracle import batch as ob
    from oracle.ppo import OraclePPO
    n, O, A, P = 64, 17, 6, 32
    torch.manual_seed(5)
    env = SyntheticVecEnv(n, O, A, P, min_len=5, seed=7)
    pol = GaussianMLPPolicy(env.spec, hidden_sizes=(32, 32))
    vf = GaussianMLPValueFunction(env.spec, hidden_sizes=(32, 32))
    with torch.no_grad():
        vf.net.params.add_(torch.randn_like(vf.net.params) * 0.05)
        for l in range(3):  # keep the layout contract: padded columns stay 0
            w = vf.net.params[vf.net.w_off[l]:vf.net.b_off[l]].view(
                vf.net.dims[l + 1], -1)
            w[:, vf.net.dims[l]:] = 0
    sampler = GpuVecSampler(pol, env, max_episode_length=P, n_workers=1,
                            worker_class=GpuVecWorker,
                            worker_args=dict(n_envs=n))
    E, mb = 3, 200
    algo = PPO(env_spec=env.spec, policy=pol, value_function=vf,
               sampler=sampler,
               policy_optimizer=OptimizerWrapper(
                   (torch.optim.Adam, dict(lr=2.5e-4)), pol,
                   max_optimization_epochs=E, minibatch_size=mb),
               vf_optimizer=OptimizerWrapper(
                   (torch.optim.Adam, dict(lr=2.5e-4)), vf,
                   max_optimization_epochs=E, minibatch_size=mb))
    oracle = OraclePPO(pol.state_dict(), vf.state_dict(), max_episode_length=P,
                       max_optimization_epochs=E, minibatch_size=mb)
    for it in range(2):
        eps = sampler.obtain_samples(it, n * P, pol.get_param_values())
        host = ob.OracleEpisodeBatch(
            observations=eps.observations,
            last_observations=eps.last_observations, actions=eps.actions,
            rewards=eps.rewards, step_types=eps.step_types,
            lengths=eps.lengths, max_episode_length=P)
        np.random.seed(50 + it)
        want = oracle.train_once(host)
        np.random.seed(50 + it)
        algo._train_once(it, eps)
        got = algo.last_tabular
        for k in LOG_KEYS:
            assert np.isclose(got[k], want[k], atol=1e-5, rtol=1e-5), (k, it)
        t = algo.last_tensors
        assert np.allclose(t['returns'].cpu().numpy(), want['returns_flat'],
                           rtol=1e-5, atol=1e-5)
        assert np.allclose(t['advantages'].cpu().numpy(),
                           want['advantages_flat'], rtol=1e-4, atol=1e-5)
        wp, wv = oracle.state()
        for k, v in pol.state_dict().items():
            assert np.allclose(v.numpy(), wp[k], atol=1e-5), k
        for k, v in vf.state_dict().items():
            assert np.allclose(v.numpy(), wv[k], atol=1e-5), k
        for k, v in algo.last_performance.items():
            assert np.isclose(v, want['performance'][k]), k


# ---------------------------------------------------------------------------
# categorical head (no torch CategoricalMLPPolicy exists in the reference,
# the tests against the oracle's torch.distributions.Categorical
# restatement come first, then the one against the reference's own
# CategoricalCNNPolicy configured as an MLP, train_once_categorical.npz)
def _discrete_spec(O, n_act, P):
    from garage_amd._dtypes import Box, Discrete, EnvSpec
    return EnvSpec(Box(-np.inf, np.inf, (O, )), Discrete(n_act),
                   max_episode_length=P)


@pytest.mark.parametrize('dbl,ent', [(True, None), (False, None),
                                     (True, (0.03, False)),
                                     (False, (0.03, True))])
def test_categorical_loss_and_grads(dbl, ent):
    from garage_amd._lib import call, dptr, stream_ptr
    from garage_amd.engine import pad_rows, reduction_workspace
    from garage_amd.policies import CategoricalMLPPolicy
    from oracle import networks as nets
    O, A, M = 5, 4, 600
    torch.manual_seed(2)
    pol = CategoricalMLPPolicy(_discrete_spec(O, A, 8), hidden_sizes=(16, 16),
                               double_softmax=dbl)
    dev = pol.device
    rng = np.random.RandomState(3)
    obs = torch.from_numpy(rng.randn(M, O).astype(np.float32))
    act = torch.from_numpy(rng.randint(0, A, M).astype(np.float32))
    adv = torch.from_numpy(rng.randn(M).astype(np.float32))
    sd = pol.state_dict()
    params = OrderedDict((k, v.clone().requires_grad_(True))
                         for k, v in sd.items())
    with torch.no_grad():
        old_ll = nets.categorical_dist(sd, '_module.', obs, dbl).log_prob(
            act.long()) + torch.from_numpy(
                (rng.randn(M) * 0.3).astype(np.float32))
    dist = nets.categorical_dist(params, '_module.', obs, dbl)
    ll = dist.log_prob(act.long())
    ratio = (ll - old_ll).exp()
    obj = torch.min(ratio * adv, torch.clamp(ratio, 0.8, 1.2) * adv)
    if ent is not None:
        e = dist.entropy()
        if ent[1]:
            e = torch.nn.functional.softplus(e)
        obj = obj + ent[0] * e
    loss = -obj.mean()
    loss.backward()

    net = pol.net
    X = pad_rows(obs)
    scores = net.forward(X, M)
    dout = net.dout_view(M)
    dout.zero_()
    ll_out = torch.empty(M, device=dev)
    loss_out = torch.zeros(1, device=dev)
    flags = 0 if ent is None else (1 | (2 if ent[1] else 0))
    actd = pad_rows(act.reshape(-1, 1))
    olld, advd = old_ll.to(dev), adv.to(dev)  # keep alive across the call
    call('ga_ppo_categorical_loss_f32', dptr(scores), scores.stride(0),
         dptr(actd), actd.stride(0), dptr(olld), dptr(advd),
         None, M, A, int(dbl), 0, 0.2, ent[0] if ent else 0.0, flags,
         dptr(dout), dptr(ll_out), None, dptr(loss_out), None,
         dptr(net._slabs), net.n_flat, int(net._splits),
         dptr(reduction_workspace(dev)), stream_ptr())
    net.backward(X, M, dout)
    net.reduce_grads()
    assert np.isclose(loss_out.item(), loss.item(), atol=1e-6, rtol=1e-5)
    assert np.allclose(ll_out.cpu().numpy(), ll.detach().numpy(), atol=1e-5)
    for key, view in net.named_views(net.grads):
        if key == '_init_std':
            assert float(view.abs().max()) == 0.0
            continue
        ref = params['_module.' + key].grad.numpy()
        got = view.cpu().numpy().reshape(ref.shape)
        assert np.allclose(got, ref, atol=2e-6 + 1e-4 * np.abs(ref).max()), key


@pytest.mark.parametrize('kw', [dict(), dict(entropy_method='max',
                                             policy_ent_coeff=0.05,
                                             center_adv=False,
                                             stop_entropy_gradient=True),
                                dict(entropy_method='regularized',
                                     policy_ent_coeff=0.02)])
def test_categorical_iteration_vs_oracle(kw):
    """Discrete synthetic envs -> GPU sampler -> PPO vs the oracle (ragged)."""
    from garage_amd.algos import PPO
    from garage_amd.envs import SyntheticVecEnv
    from garage_amd.optimizers import OptimizerWrapper
    from garage_amd.policies import (CategoricalMLPPolicy,
                                     GaussianMLPValueFunction)
    from garage_amd.sampler import GpuVecSampler, GpuVecWorker
    from oracle import batch as ob
    from oracle.ppo import OraclePPO
    n, O, A, P = 48, 4, 2, 20
    torch.manual_seed(9)
    env = SyntheticVecEnv(n, O, A, P, min_len=4, seed=11, discrete=True)
    pol = CategoricalMLPPolicy(env.spec, hidden_sizes=(16, 16))
    vf = GaussianMLPValueFunction(env.spec, hidden_sizes=(16, 16))
    sampler = GpuVecSampler(pol, env, max_episode_length=P, n_workers=1,
                            worker_class=GpuVecWorker,
                            worker_args=dict(n_envs=n))
    E, mb = 2, 128
    opt = (torch.optim.Adam, dict(lr=1e-3))
    algo = PPO(env_spec=env.spec, policy=pol, value_function=vf,
               sampler=sampler,
               policy_optimizer=OptimizerWrapper(opt, pol, E, mb),
               vf_optimizer=OptimizerWrapper(opt, vf, E, mb), **kw)
    oracle = OraclePPO(pol.state_dict(), vf.state_dict(), max_episode_length=P,
                       policy_kind='categorical', max_optimization_epochs=E,
                       minibatch_size=mb, policy_lr=1e-3, vf_lr=1e-3, **kw)
    for it in range(2):
        eps = sampler.obtain_samples(it, n * P, None)
        assert eps.actions.dtype == np.int64 and eps.actions.ndim == 1
        assert set(np.unique(eps.actions)) <= {0, 1}
        host = ob.OracleEpisodeBatch(
            observations=eps.observations,
            last_observations=eps.last_observations, actions=eps.actions,
            rewards=eps.rewards, step_types=eps.step_types,
            lengths=eps.lengths, max_episode_length=P)
        np.random.seed(70 + it)
        want = oracle.train_once(host)
        np.random.seed(70 + it)
        algo._train_once(it, eps)
        for k in LOG_KEYS:
            assert np.isclose(algo.last_tabular[k], want[k], atol=1e-5,
                              rtol=1e-5), (k, it, algo.last_tabular[k],
                                           want[k])
        wp, wv = oracle.state()
        for k, v in pol.state_dict().items():
            assert np.allclose(v.numpy(), wp[k], atol=1e-5), k
        for k, v in vf.state_dict().items():
            assert np.allclose(v.numpy(), wv[k], atol=1e-5), k


@pytest.mark.parametrize('tag', sorted(__import__('_categorical_golden').CASES))
def test_categorical_train_once_matches_real_reference(golden, tag):
    """The categorical head against the REAL reference (not only the oracle):
    two ``_train_once`` iterations of the real PPO / VPG on the reference's torch
    categorical policy, ``CategoricalCNNPolicy`` configured as an MLP (a 1 x 1
    convolution of a 1 x 1 image is a dense layer; tests/_categorical_golden.py,
    ``categorical_cnn_policy.py:115-140``): forward distribution, 9 logged
    scalars, post-update parameters and Adam moments; ``ppo_c2`` has C2's
    widths (obs 4, 2 actions, MLP(64, 64))."""
    import _categorical_golden as cg
    from garage_amd.algos import PPO, VPG
    from garage_amd.optimizers import OptimizerWrapper
    from garage_amd.policies import (CategoricalMLPPolicy,
                                     GaussianMLPValueFunction)
    g = golden('train_once_categorical')
    O, n_act, P, E, mb = [int(v) for v in g[tag + '_cfg']]
    mb = None if mb < 0 else mb
    hidden = tuple(int(h) for h in g[tag + '_hidden'])
    spec = _discrete_spec(O, n_act, P)
    pol = CategoricalMLPPolicy(spec, hidden_sizes=hidden)
    vf = GaussianMLPValueFunction(spec, hidden_sizes=hidden)
    pol.load_state_dict(cg.policy_params(g, tag + '_pol0:'))
    vf.load_state_dict(_sd(g, tag + '_vf0:'))
    with torch.no_grad():
        dist, info = pol(torch.from_numpy(g[tag + '_fwd_obs']))
        act = torch.from_numpy(g[tag + '_fwd_act']).to(dist.probs.device)
        assert info == {}
        assert np.allclose(dist.probs.cpu(), g[tag + '_fwd_probs'], atol=1e-6)
        assert np.allclose(dist.log_prob(act).cpu(), g[tag + '_fwd_log_prob'],
                           atol=2e-6)
        assert np.allclose(dist.entropy().cpu(), g[tag + '_fwd_entropy'],
                           atol=2e-6)
    cls = VPG if tag == 'vpg' else PPO
    algo = cls(env_spec=spec, policy=pol, value_function=vf, sampler=None,
               policy_optimizer=OptimizerWrapper(
                   (torch.optim.Adam, dict(lr=1e-3)), pol,
                   max_optimization_epochs=E, minibatch_size=mb),
               vf_optimizer=OptimizerWrapper(
                   (torch.optim.Adam, dict(lr=1e-3)), vf,
                   max_optimization_epochs=E, minibatch_size=mb),
               **cg.CASES[tag])
    for it in range(2):
        pre = '%s_it%d_' % (tag, it)
        batch = _host_batch(spec, g, pre, O)
        assert batch.actions.dtype == np.int64 and batch.actions.ndim == 1
        np.random.seed(int(g[pre + 'np_seed']))
        avg = algo._train_once(it, batch)
        for mine, theirs in cg.LOG_KEYS.items():
            want = float(g[pre + 'log:' + theirs])
            assert np.isclose(algo.last_tabular[mine], want, atol=1e-5,
                              rtol=1e-5), (mine, it, algo.last_tabular[mine],
                                           want)
        assert np.isclose(avg, float(g[pre + 'avg_return']))
        for k, v in algo.last_performance.items():
            assert np.isclose(v, float(g[pre + 'log:Evaluation/' + k])), k
        want_pol = cg.policy_params(g, pre + 'pol:')
        assert sorted(pol.state_dict()) == sorted(want_pol)
        for k, v in pol.state_dict().items():
            assert np.allclose(v.numpy(), want_pol[k].numpy(), atol=2e-6), k
        for k, v in vf.state_dict().items():
            assert np.allclose(v.numpy(), g[pre + 'vf:' + k], atol=2e-6), k
        for name, net in (('pol', pol.net), ('vf', vf.net)):
            # (named_views starts with the std slot, which this head lacks)
            views_m = [mv for mv in net.named_views(net.exp_avg)
                       if name == 'vf' or mv[0] != '_init_std']
            views_v = [mv for mv in net.named_views(net.exp_avg_sq)
                       if name == 'vf' or mv[0] != '_init_std']
            for j, ((_, m), (_, v)) in enumerate(zip(views_m, views_v)):
                gm = cg.flat(g['%sadam_%s_%d_m' % (pre, name, j)])
                gv = cg.flat(g['%sadam_%s_%d_v' % (pre, name, j)])
                assert np.allclose(m.cpu().numpy().reshape(gm.shape), gm,
                                   atol=1e-6), (name, j)
                assert np.allclose(v.cpu().numpy().reshape(gv.shape), gv,
                                   atol=1e-8), (name, j)
            assert net.adam_steps == int(g['%sadam_%s_0_step' % (pre, name)])


def test_opt_in_fused_head_loss_iteration():
    """``fuse_head = True`` (head layer computed inside the loss kernel): the
    native epoch loop and the Python minibatch loop give the same bits, and
    both agree with the default head GEMM + loss kernel pair to rounding."""
    from garage_amd._dtypes import EpisodeBatch, StepType
    from garage_amd.algos import PPO
    from garage_amd.optimizers import OptimizerWrapper
    from garage_amd.policies import GaussianMLPPolicy, GaussianMLPValueFunction
    O, A, P = 6, 3, 12
    spec = _spec(O, A, P)
    rng = np.random.RandomState(1)
    lens = rng.randint(4, P + 1, size=40)
    lens[0] = P
    S = int(lens.sum())
    st = []
    for L in lens:
        t = [1] * L
        t[0] = 0
        t[-1] = 3 if L == P else 2
        st += t
    batch = EpisodeBatch(
        env_spec=spec, episode_infos={},
        observations=rng.randn(S, O).astype(np.float32),
        last_observations=np.zeros((len(lens), O), np.float32),
        actions=rng.randn(S, A).astype(np.float32), rewards=rng.randn(S),
        env_infos={}, agent_infos={},
        step_types=np.asarray([StepType(s) for s in st], dtype=object),
        lengths=lens.astype('l'))

    class LoopPPO(PPO):  # overriding a per-minibatch hook forces the Python loop

        def _train_policy(self, *a):
            return super()._train_policy(*a)

    def run(cls, fuse):
        torch.manual_seed(2)
        pol = GaussianMLPPolicy(spec, hidden_sizes=(64, 64))
        vf = GaussianMLPValueFunction(spec, hidden_sizes=(64, 128))
        opt = (torch.optim.Adam, dict(lr=1e-3))
        algo = cls(env_spec=spec, policy=pol, value_function=vf, sampler=None,
                   policy_optimizer=OptimizerWrapper(opt, pol, 2, 64,
                                                     permutation='device',
                                                     seed=3),
                   vf_optimizer=OptimizerWrapper(opt, vf, 2, 64,
                                                 permutation='device', seed=4))
        algo.fuse_head = fuse
        assert pol.net.head_fusable() and vf.net.head_fusable()
        algo._train_once(0, batch)
        return pol.net.params.clone(), vf.net.params.clone(), \
            dict(algo.last_tabular)

    base = run(PPO, False)
    fused = run(PPO, True)
    loop = run(LoopPPO, True)
    assert torch.equal(fused[0], loop[0]) and torch.equal(fused[1], loop[1])
    assert torch.allclose(base[0], fused[0], atol=2e-6)
    assert torch.allclose(base[1], fused[1], atol=2e-6)
    for k in base[2]:
        assert np.isclose(base[2][k], fused[2][k], atol=1e-5, rtol=1e-5), k


@pytest.mark.parametrize('algo_name', ['vpg', 'ppo', 'trpo'])
def test_reference_named_evaluation_helpers(algo_name):
    """``_compute_objective`` / ``_compute_loss_with_adv`` / ``_compute_loss`` /
    ``_compute_kl_constraint`` / ``_compute_policy_entropy`` (vpg.py:295-455,
    ppo.py:96-132, trpo.py:93-119) against the oracle's restatements, with the
    current policy moved away from the old one."""
    from garage_amd.algos import PPO, TRPO, VPG
    from garage_amd.policies import GaussianMLPPolicy, GaussianMLPValueFunction
    from oracle.ppo import OraclePPO
    from oracle.trpo import OracleTRPO
    O, A, P, N = 5, 3, 7, 9
    spec = _spec(O, A, P)
    torch.manual_seed(8)
    rng = np.random.RandomState(8)
    pol = GaussianMLPPolicy(spec, hidden_sizes=(16, 16))
    vf = GaussianMLPValueFunction(spec, hidden_sizes=(16, 16))
    cls = dict(vpg=VPG, ppo=PPO, trpo=TRPO)[algo_name]
    kw = dict(entropy_method='regularized', policy_ent_coeff=0.03)
    algo = cls(env_spec=spec, policy=pol, value_function=vf, sampler=None, **kw)
    old_sd = OrderedDict((k, v.clone()) for k, v in pol.state_dict().items())
    ocls = OracleTRPO if algo_name == 'trpo' else OraclePPO
    okw = dict(kw)
    if algo_name != 'trpo':
        okw['algo'] = algo_name
    oracle = ocls(old_sd, OrderedDict(vf.state_dict()), max_episode_length=P,
                  gae_lambda=algo._gae_lambda, **okw)
    # move the current policy (the old one stays where it was)
    new_sd = OrderedDict((k, v + 0.05 * torch.randn_like(v)
                          if 'min_std' not in k else v)
                         for k, v in old_sd.items())
    pol.load_state_dict(new_sd)
    for k in oracle.policy:
        oracle.policy[k].data.copy_(new_sd[k])
    lens = rng.randint(2, P + 1, size=N)
    S = int(lens.sum())
    obs = rng.randn(S, O).astype(np.float32)
    act = rng.randn(S, A).astype(np.float32)
    adv = rng.randn(S).astype(np.float32)
    tobs, tact, tadv = (torch.from_numpy(obs), torch.from_numpy(act),
                        torch.from_numpy(adv))
    with torch.no_grad():
        want_obj = oracle._objective(tadv, tobs, tact)
        want_loss = oracle._policy_loss(tobs, tact, tadv)
        want_kl = oracle._kl(tobs)
        want_ent = oracle._entropy(tobs)
    got_obj = algo._compute_objective(tadv, tobs, tact, None).cpu()
    assert torch.allclose(got_obj, want_obj, atol=2e-5, rtol=2e-5)
    got_loss = algo._compute_loss_with_adv(tobs, tact, None, tadv).cpu()
    assert np.isclose(float(got_loss), float(want_loss), atol=1e-5, rtol=1e-5)
    got_kl = algo._compute_kl_constraint(tobs).cpu()
    assert np.isclose(float(got_kl), float(want_kl), atol=1e-6, rtol=1e-4)
    got_ent = algo._compute_policy_entropy(tobs).cpu()
    assert got_ent.shape == want_ent.shape
    assert torch.allclose(got_ent, want_ent, atol=1e-5)
    # padded entry point: (N, P, ...) inputs + valids + baselines
    from oracle.returns import vpg_compute_advantage
    pobs = np.zeros((N, P, O), np.float32)
    pact = np.zeros((N, P, A), np.float32)
    prew = np.zeros((N, P), np.float32)
    base = rng.randn(N, P).astype(np.float32)
    off = 0
    for i, L in enumerate(lens):
        pobs[i, :L], pact[i, :L] = obs[off:off + L], act[off:off + L]
        prew[i, :L] = rng.randn(L)
        off += L
    got = algo._compute_loss(pobs, pact, prew, lens, base).cpu()
    with torch.no_grad():
        adv_ref = vpg_compute_advantage(algo._discount, algo._gae_lambda, P,
                                        torch.from_numpy(prew), lens,
                                        torch.from_numpy(base), True, False)
        want = oracle._policy_loss(tobs, tact, adv_ref)
    assert np.isclose(float(got), float(want), atol=2e-5, rtol=1e-4)


POLICY_OPTIONS = {
    'fixed_std': dict(learn_std=False, init_std=0.7),
    'max_clamp': dict(max_std=0.5, init_std=1.0),
    'min_clamp': dict(min_std=0.2, init_std=0.1),
    'init_small': dict(init_std=0.3),
}


@pytest.mark.parametrize('tag', sorted(POLICY_OPTIONS))
def test_policy_std_options_match_real_reference(golden, tag):
    """``GaussianMLPPolicy(learn_std / init_std / min_std / max_std)`` through two
    real PPO iterations (tests/golden/policy_options.npz): fixed std is a buffer
    and never moves, an active clamp passes no gradient, state_dict keys follow
    the reference (``init_std`` vs ``_init_std``, ``max_std_param``)."""
    from garage_amd.algos import PPO
    from garage_amd.optimizers import OptimizerWrapper
    from garage_amd.policies import GaussianMLPPolicy, GaussianMLPValueFunction
    g = golden('policy_options')
    O, A, P, E, mb = [int(v) for v in g[tag + '_cfg']]
    spec = _spec(O, A, P)
    pol = GaussianMLPPolicy(spec, hidden_sizes=(8, 8), **POLICY_OPTIONS[tag])
    vf = GaussianMLPValueFunction(spec, hidden_sizes=(8, 8))
    want_keys = sorted(k[len(tag + '_pol0:'):] for k in g.files
                       if k.startswith(tag + '_pol0:'))
    assert sorted(pol.state_dict().keys()) == want_keys
    pol.load_state_dict(_sd(g, tag + '_pol0:'))
    vf.load_state_dict(_sd(g, tag + '_vf0:'))
    algo = PPO(env_spec=spec, policy=pol, value_function=vf, sampler=None,
               policy_optimizer=OptimizerWrapper(
                   (torch.optim.Adam, dict(lr=2.5e-3)), pol,
                   max_optimization_epochs=E, minibatch_size=mb),
               vf_optimizer=OptimizerWrapper(
                   (torch.optim.Adam, dict(lr=2.5e-3)), vf,
                   max_optimization_epochs=E, minibatch_size=mb))
    for it in range(2):
        pre = '%s_it%d_' % (tag, it)
        batch = _host_batch(spec, g, pre, O)
        np.random.seed(int(g[pre + 'np_seed']))
        algo._train_once(it, batch)
        for mine, theirs in LOG_KEYS.items():
            want = float(g[pre + 'log:' + theirs])
            assert np.isclose(algo.last_tabular[mine], want, atol=1e-5,
                              rtol=1e-5), (mine, it, algo.last_tabular[mine],
                                           want)
        for k, v in pol.state_dict().items():
            assert np.allclose(v.numpy(), g[pre + 'pol:' + k], atol=2e-6), k
        for k, v in vf.state_dict().items():
            assert np.allclose(v.numpy(), g[pre + 'vf:' + k], atol=2e-6), k


ACTIVATION_CASES = {
    # tag: (policy hidden_nonlinearity, value-function hidden_nonlinearity)
    'relu': (torch.relu, torch.relu),
    'linear': (None, None),
    'relu_policy_tanh_vf': (torch.nn.ReLU, torch.tanh),
}


@pytest.mark.parametrize('tag', sorted(ACTIVATION_CASES))
def test_hidden_nonlinearities_match_real_reference(golden, tag):
    """``hidden_nonlinearity`` = relu / None (``torch/modules/mlp_module.py:43-44``):
    forward outputs of the real networks and two real PPO iterations
    (tests/golden/policy_options.npz).  Such networks take the per-layer GEMM
    kernels (the fused / one-launch kernels implement tanh)."""
    from garage_amd.algos import PPO
    from garage_amd.engine import pad_rows
    from garage_amd.optimizers import OptimizerWrapper
    from garage_amd.policies import GaussianMLPPolicy, GaussianMLPValueFunction
    g = golden('policy_options')
    O, A, P, E, mb = [int(v) for v in g[tag + '_cfg']]
    spec = _spec(O, A, P)
    pa, va = ACTIVATION_CASES[tag]
    pol = GaussianMLPPolicy(spec, hidden_sizes=(8, 8), hidden_nonlinearity=pa)
    vf = GaussianMLPValueFunction(spec, hidden_sizes=(8, 8),
                                  hidden_nonlinearity=va)
    pol.load_state_dict(_sd(g, tag + '_pol0:'))
    vf.load_state_dict(_sd(g, tag + '_vf0:'))
    x = torch.from_numpy(g[tag + '_fwd_obs'])
    dist, _ = pol.forward(x)
    assert np.allclose(dist.mean.cpu().numpy(), g[tag + '_fwd_mean'], atol=2e-6)
    assert np.allclose(vf.forward(x).cpu().numpy().reshape(-1),
                       g[tag + '_fwd_value'].reshape(-1), atol=2e-6)
    algo = PPO(env_spec=spec, policy=pol, value_function=vf, sampler=None,
               policy_optimizer=OptimizerWrapper(
                   (torch.optim.Adam, dict(lr=2.5e-3)), pol,
                   max_optimization_epochs=E, minibatch_size=mb),
               vf_optimizer=OptimizerWrapper(
                   (torch.optim.Adam, dict(lr=2.5e-3)), vf,
                   max_optimization_epochs=E, minibatch_size=mb))
    for it in range(2):
        pre = '%s_it%d_' % (tag, it)
        batch = _host_batch(spec, g, pre, O)
        np.random.seed(int(g[pre + 'np_seed']))
        algo._train_once(it, batch)
        for mine, theirs in LOG_KEYS.items():
            want = float(g[pre + 'log:' + theirs])
            assert np.isclose(algo.last_tabular[mine], want, atol=1e-5,
                              rtol=1e-5), (mine, it, algo.last_tabular[mine],
                                           want)
        for k, v in pol.state_dict().items():
            assert np.allclose(v.numpy(), g[pre + 'pol:' + k], atol=2e-6), k
        for k, v in vf.state_dict().items():
            assert np.allclose(v.numpy(), g[pre + 'vf:' + k], atol=2e-6), k


OUTPUT_CASES = {
    # tag: (policy kwargs, value-function kwargs)
    'out_tanh': (dict(output_nonlinearity=torch.tanh), dict()),
    'out_tanh_vf_relu_hidden': (dict(output_nonlinearity=torch.tanh,
                                     hidden_nonlinearity=torch.relu),
                                dict(output_nonlinearity=torch.tanh)),
}


@pytest.mark.parametrize('tag', sorted(OUTPUT_CASES))
def test_output_nonlinearity_matches_real_reference(golden, tag):
    """``output_nonlinearity`` on the Gaussian mean / the value
    (``torch/modules/mlp_module.py:52-53``): forward outputs of the real networks
    and two real PPO iterations (tests/golden/policy_options.npz); the loss's
    gradient is scaled by the slope at the output before the backward pass."""
    from garage_amd.algos import PPO
    from garage_amd.optimizers import OptimizerWrapper
    from garage_amd.policies import GaussianMLPPolicy, GaussianMLPValueFunction
    g = golden('policy_options')
    O, A, P, E, mb = [int(v) for v in g[tag + '_cfg']]
    spec = _spec(O, A, P)
    pkw, vkw = OUTPUT_CASES[tag]
    pol = GaussianMLPPolicy(spec, hidden_sizes=(8, 8), **pkw)
    vf = GaussianMLPValueFunction(spec, hidden_sizes=(8, 8), **vkw)
    pol.load_state_dict(_sd(g, tag + '_pol0:'))
    vf.load_state_dict(_sd(g, tag + '_vf0:'))
    x = torch.from_numpy(g[tag + '_fwd_obs'])
    dist, _ = pol.forward(x)
    assert np.allclose(dist.mean.cpu().numpy(), g[tag + '_fwd_mean'], atol=2e-6)
    assert np.allclose(vf.forward(x).cpu().numpy().reshape(-1),
                       g[tag + '_fwd_value'].reshape(-1), atol=2e-6)
    algo = PPO(env_spec=spec, policy=pol, value_function=vf, sampler=None,
               policy_optimizer=OptimizerWrapper(
                   (torch.optim.Adam, dict(lr=2.5e-3)), pol,
                   max_optimization_epochs=E, minibatch_size=mb),
               vf_optimizer=OptimizerWrapper(
                   (torch.optim.Adam, dict(lr=2.5e-3)), vf,
                   max_optimization_epochs=E, minibatch_size=mb))
    for it in range(2):
        pre = '%s_it%d_' % (tag, it)
        batch = _host_batch(spec, g, pre, O)
        np.random.seed(int(g[pre + 'np_seed']))
        algo._train_once(it, batch)
        for mine, theirs in LOG_KEYS.items():
            want = float(g[pre + 'log:' + theirs])
            assert np.isclose(algo.last_tabular[mine], want, atol=1e-5,
                              rtol=1e-5), (mine, it, algo.last_tabular[mine],
                                           want)
        for k, v in pol.state_dict().items():
            assert np.allclose(v.numpy(), g[pre + 'pol:' + k], atol=2e-6), k
        for k, v in vf.state_dict().items():
            assert np.allclose(v.numpy(), g[pre + 'vf:' + k], atol=2e-6), k


# round 3 (tests/golden/policy_activations.npz): tag -> (policy kwargs, vf kwargs),
# spelled differently from the generator where the reference accepts several forms
MORE_ACTIVATION_CASES = {
    'sigmoid': (dict(hidden_nonlinearity=torch.nn.Sigmoid),
                dict(hidden_nonlinearity=torch.sigmoid)),
    'elu': (dict(hidden_nonlinearity=torch.nn.functional.elu),
            dict(hidden_nonlinearity=torch.nn.ELU())),
    'leaky_relu': (dict(hidden_nonlinearity=torch.nn.LeakyReLU),
                   dict(hidden_nonlinearity=torch.nn.functional.leaky_relu)),
    'softplus_hidden': (dict(hidden_nonlinearity=torch.nn.functional.softplus),
                        dict(hidden_nonlinearity=torch.nn.Softplus)),
    'out_sigmoid_elu_hidden': (
        dict(output_nonlinearity=torch.sigmoid,
             hidden_nonlinearity=torch.nn.functional.elu),
        dict(output_nonlinearity=torch.nn.functional.softplus)),
}


@pytest.mark.parametrize('tag', sorted(MORE_ACTIVATION_CASES))
def test_more_nonlinearities_match_real_reference(golden, tag):
    """sigmoid / elu / leaky_relu / softplus as hidden or output nonlinearity
    (``NonLinearity``, ``torch/modules/multi_headed_mlp_module.py:154-197``):
    forward outputs of the real networks and two real PPO iterations.  Their
    slopes are functions of the OUTPUT (h (1 - h), h + 1, 0.01, 1 - exp(-h)), so
    the backward pass reads the same activations as for tanh."""
    from garage_amd.algos import PPO
    from garage_amd.optimizers import OptimizerWrapper
    from garage_amd.policies import GaussianMLPPolicy, GaussianMLPValueFunction
    g = golden('policy_activations')
    O, A, P, E, mb = [int(v) for v in g[tag + '_cfg']]
    spec = _spec(O, A, P)
    pkw, vkw = MORE_ACTIVATION_CASES[tag]
    pol = GaussianMLPPolicy(spec, hidden_sizes=(8, 8), **pkw)
    vf = GaussianMLPValueFunction(spec, hidden_sizes=(8, 8), **vkw)
    pol.load_state_dict(_sd(g, tag + '_pol0:'))
    vf.load_state_dict(_sd(g, tag + '_vf0:'))
    x = torch.from_numpy(g[tag + '_fwd_obs'])
    dist, _ = pol.forward(x)
    assert np.allclose(dist.mean.cpu().numpy(), g[tag + '_fwd_mean'], atol=2e-6)
    assert np.allclose(vf.forward(x).cpu().numpy().reshape(-1),
                       g[tag + '_fwd_value'].reshape(-1), atol=2e-6)
    algo = PPO(env_spec=spec, policy=pol, value_function=vf, sampler=None,
               policy_optimizer=OptimizerWrapper(
                   (torch.optim.Adam, dict(lr=2.5e-3)), pol,
                   max_optimization_epochs=E, minibatch_size=mb),
               vf_optimizer=OptimizerWrapper(
                   (torch.optim.Adam, dict(lr=2.5e-3)), vf,
                   max_optimization_epochs=E, minibatch_size=mb))
    for it in range(2):
        pre = '%s_it%d_' % (tag, it)
        batch = _host_batch(spec, g, pre, O)
        np.random.seed(int(g[pre + 'np_seed']))
        algo._train_once(it, batch)
        for mine, theirs in LOG_KEYS.items():
            want = float(g[pre + 'log:' + theirs])
            assert np.isclose(algo.last_tabular[mine], want, atol=1e-5,
                              rtol=1e-5), (mine, it, algo.last_tabular[mine],
                                           want)
        for k, v in pol.state_dict().items():
            assert np.allclose(v.numpy(), g[pre + 'pol:' + k], atol=2e-6), k
        for k, v in vf.state_dict().items():
            assert np.allclose(v.numpy(), g[pre + 'vf:' + k], atol=2e-6), k


def test_unsupported_nonlinearities_are_refused():
    """Parameterised activations away from torch's defaults, and callables whose
    slope is not a function of their output, raise instead of silently computing
    something else."""
    from garage_amd.policies import GaussianMLPPolicy
    spec = _spec(4, 2, 8)
    for bad in (torch.nn.ELU(alpha=0.5), torch.nn.LeakyReLU(0.2),
                torch.nn.Softplus(beta=2), torch.nn.functional.gelu, torch.sin):
        with pytest.raises(NotImplementedError):
            GaussianMLPPolicy(spec, hidden_sizes=(8, 8),
                              hidden_nonlinearity=bad)


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
def test_other_optimizers_match_real_reference(golden, tag):
    """``make_optimizer`` (``_functions.py:25-65``) builds any torch.optim class:
    SGD (momentum / dampening / nesterov / weight decay), RMSprop (centred,
    momentum) and Adam / AdamW with weight decay or amsgrad step through
    ``ga_optimizer_step_f32`` behind the per-minibatch loop -- two real PPO
    iterations of the reference each (tests/golden/train_once_optimizers.npz),
    then a pickle round trip that must carry the optimizer state."""
    import pickle

    from garage_amd.algos import PPO
    from garage_amd.optimizers import OptimizerWrapper
    from garage_amd.policies import GaussianMLPPolicy, GaussianMLPValueFunction
    g = golden('train_once_optimizers')
    O, A, P, E, mb = [int(v) for v in g[tag + '_cfg']]
    spec = _spec(O, A, P)
    cls, kw = OPTIMIZER_CASES[tag]
    pol = GaussianMLPPolicy(spec, hidden_sizes=(8, 8))
    vf = GaussianMLPValueFunction(spec, hidden_sizes=(8, 8))
    pol.load_state_dict(_sd(g, tag + '_pol0:'))
    vf.load_state_dict(_sd(g, tag + '_vf0:'))
    algo = PPO(env_spec=spec, policy=pol, value_function=vf, sampler=None,
               policy_optimizer=OptimizerWrapper((cls, dict(kw)), pol,
                                                 max_optimization_epochs=E,
                                                 minibatch_size=mb),
               vf_optimizer=OptimizerWrapper((cls, dict(kw)), vf,
                                             max_optimization_epochs=E,
                                             minibatch_size=mb))
    assert not algo._native_update_ok()
    for it in range(2):
        if it == 1:  # the second iteration runs on a restored copy
            algo = pickle.loads(pickle.dumps(algo))
            pol, vf = algo.policy, algo._value_function
        pre = '%s_it%d_' % (tag, it)
        batch = _host_batch(spec, g, pre, O)
        np.random.seed(int(g[pre + 'np_seed']))
        algo._train_once(it, batch)
        for mine, theirs in LOG_KEYS.items():
            want = float(g[pre + 'log:' + theirs])
            assert np.isclose(algo.last_tabular[mine], want, atol=1e-5,
                              rtol=1e-5), (mine, it, algo.last_tabular[mine],
                                           want)
        for k, v in pol.state_dict().items():
            assert np.allclose(v.numpy(), g[pre + 'pol:' + k], atol=2e-6), k
        for k, v in vf.state_dict().items():
            assert np.allclose(v.numpy(), g[pre + 'vf:' + k], atol=2e-6), k


def test_output_nonlinearity_python_loop_and_trpo_against_oracle():
    """The same option through the Python minibatch loop (``engine.backward``
    scales d(output)) and through TRPO's Fisher-vector product (tangent and seed
    both pass the output slope): one iteration each against the oracle."""
    from garage_amd._dtypes import EpisodeBatch, StepType
    from garage_amd.algos import PPO, TRPO
    from garage_amd.optimizers import (ConjugateGradientOptimizer,
                                       OptimizerWrapper)
    from garage_amd.policies import GaussianMLPPolicy, GaussianMLPValueFunction
    from oracle import batch as ob
    from oracle import networks as nets
    from oracle.ppo import OraclePPO
    from oracle.trpo import OracleTRPO
    O, A, P = 7, 3, 16
    spec = _spec(O, A, P)
    rng = np.random.RandomState(21)
    lens = rng.randint(3, P + 1, size=40)
    lens[::4] = P
    S = int(lens.sum())
    st = []
    for L in lens:
        t = [1] * L
        t[0] = 0
        t[-1] = 3 if L == P else 2
        st += t
    obs = rng.randn(S, O).astype(np.float32)
    acts = (0.5 * rng.randn(S, A)).astype(np.float32)
    rew = rng.randn(S)

    def batches():
        b = ob.OracleEpisodeBatch(
            observations=obs, last_observations=np.zeros((len(lens), O),
                                                         np.float32),
            actions=acts, rewards=rew, step_types=np.asarray(st), lengths=lens,
            max_episode_length=P)
        e = EpisodeBatch(env_spec=spec, episode_infos={}, observations=obs,
                         last_observations=np.zeros((len(lens), O), np.float32),
                         actions=acts, rewards=rew, env_infos={}, agent_infos={},
                         step_types=np.asarray([StepType(s) for s in st],
                                               dtype=object),
                         lengths=lens.astype('l'))
        return b, e

    for name in ('ppo_python_loop', 'trpo'):
        torch.manual_seed(31)
        pol = GaussianMLPPolicy(spec, hidden_sizes=(32, 32),
                                output_nonlinearity=torch.tanh)
        vf = GaussianMLPValueFunction(spec, hidden_sizes=(32, 32))
        b, e = batches()
        with nets.output_nonlinearity(policy=torch.tanh):
            if name == 'trpo':
                oracle = OracleTRPO(OrderedDict(pol.state_dict()),
                                    OrderedDict(vf.state_dict()),
                                    max_episode_length=P,
                                    max_optimization_epochs=2, minibatch_size=128)
            else:
                oracle = OraclePPO(OrderedDict(pol.state_dict()),
                                   OrderedDict(vf.state_dict()),
                                   max_episode_length=P,
                                   max_optimization_epochs=2, minibatch_size=128,
                                   policy_lr=1e-3, vf_lr=1e-3)
            np.random.seed(6)
            want = oracle.train_once(b)
            wpol, _ = oracle.state()
        if name == 'trpo':
            algo = TRPO(env_spec=spec, policy=pol, value_function=vf, sampler=None,
                        policy_optimizer=OptimizerWrapper(
                            (ConjugateGradientOptimizer,
                             dict(max_constraint_value=0.01)), pol),
                        vf_optimizer=OptimizerWrapper(
                            (torch.optim.Adam, dict(lr=2.5e-4)), vf,
                            max_optimization_epochs=2, minibatch_size=128))
        else:
            algo = PPO(env_spec=spec, policy=pol, value_function=vf, sampler=None,
                       policy_optimizer=OptimizerWrapper(
                           (torch.optim.Adam, dict(lr=1e-3)), pol,
                           max_optimization_epochs=2, minibatch_size=128),
                       vf_optimizer=OptimizerWrapper(
                           (torch.optim.Adam, dict(lr=1e-3)), vf,
                           max_optimization_epochs=2, minibatch_size=128))
            # a subclass hooking _train_policy forces the Python minibatch loop
            algo.__class__ = type('Hooked', (PPO, ), {
                '_train_policy': lambda self, *a: PPO._train_policy(self, *a)})
            assert not algo._native_update_ok()
        np.random.seed(6)
        algo._train_once(0, e)
        tol = 2e-3 if name == 'trpo' else 2e-5
        for k in ('policy/LossBefore', 'policy/LossAfter', 'policy/KL'):
            assert np.isclose(algo.last_tabular[k], want[k], atol=tol,
                              rtol=1e-3), (name, k, algo.last_tabular[k], want[k])
        scale = max(1e-3, max(float(np.abs(np.asarray(v)).max())
                              for v in wpol.values()))
        for k, v in pol.state_dict().items():
            assert np.allclose(v.numpy(), np.asarray(wpol[k]),
                               atol=(2e-3 if name == 'trpo' else 1e-5) * scale), (
                                   name, k)


LAYER_NORM_CASES = {
    'layer_norm': (dict(layer_normalization=True), dict(layer_normalization=True)),
    'layer_norm_relu': (dict(layer_normalization=True,
                             hidden_nonlinearity=torch.relu), dict()),
}


@pytest.mark.parametrize('tag', sorted(LAYER_NORM_CASES))
def test_layer_normalization_matches_real_reference(golden, tag):
    """``layer_normalization=True`` (``multi_headed_mlp_module.py:77-81``): the
    state_dict carries the LayerNorm weights in the reference's order, forward
    outputs and two real PPO iterations match, gamma / beta included."""
    from garage_amd.algos import PPO
    from garage_amd.optimizers import OptimizerWrapper
    from garage_amd.policies import GaussianMLPPolicy, GaussianMLPValueFunction
    g = golden('policy_options')
    O, A, P, E, mb = [int(v) for v in g[tag + '_cfg']]
    spec = _spec(O, A, P)
    pkw, vkw = LAYER_NORM_CASES[tag]
    torch.manual_seed(17)
    pol = GaussianMLPPolicy(spec, hidden_sizes=(8, 8), **pkw)
    vf = GaussianMLPValueFunction(spec, hidden_sizes=(8, 8), **vkw)
    want_keys = [k[len(tag + '_pol0:'):] for k in g.files
                 if k.startswith(tag + '_pol0:')]
    assert sorted(pol.state_dict().keys()) == sorted(want_keys)
    # parameters() order of the reference: LayerNorm before its Linear
    mine = [n for n, _ in pol.named_parameters()]
    theirs = [k for k in want_keys if not k.endswith('min_std_param')]
    assert mine == theirs
    # freshly built: gamma 1, beta 0, the same Linear init as the reference
    for k, v in pol.state_dict().items():
        assert np.allclose(v.numpy(), g[tag + '_pol0:' + k], atol=1e-7), k
    pol.load_state_dict(_sd(g, tag + '_pol0:'))
    vf.load_state_dict(_sd(g, tag + '_vf0:'))
    x = torch.from_numpy(g[tag + '_fwd_obs'])
    dist, _ = pol.forward(x)
    assert np.allclose(dist.mean.cpu().numpy(), g[tag + '_fwd_mean'], atol=2e-6)
    assert np.allclose(vf.forward(x).cpu().numpy().reshape(-1),
                       g[tag + '_fwd_value'].reshape(-1), atol=2e-6)
    algo = PPO(env_spec=spec, policy=pol, value_function=vf, sampler=None,
               policy_optimizer=OptimizerWrapper(
                   (torch.optim.Adam, dict(lr=2.5e-3)), pol,
                   max_optimization_epochs=E, minibatch_size=mb),
               vf_optimizer=OptimizerWrapper(
                   (torch.optim.Adam, dict(lr=2.5e-3)), vf,
                   max_optimization_epochs=E, minibatch_size=mb))
    for it in range(2):
        pre = '%s_it%d_' % (tag, it)
        batch = _host_batch(spec, g, pre, O)
        np.random.seed(int(g[pre + 'np_seed']))
        algo._train_once(it, batch)
        for mine_k, theirs_k in LOG_KEYS.items():
            want = float(g[pre + 'log:' + theirs_k])
            assert np.isclose(algo.last_tabular[mine_k], want, atol=1e-5,
                              rtol=1e-5), (mine_k, it, algo.last_tabular[mine_k],
                                           want)
        for k, v in pol.state_dict().items():
            assert np.allclose(v.numpy(), g[pre + 'pol:' + k], atol=2e-6), k
        for k, v in vf.state_dict().items():
            assert np.allclose(v.numpy(), g[pre + 'vf:' + k], atol=2e-6), k


def test_layer_normalization_at_c3_shape_against_oracle():
    """LayerNorm rows of 17 and 256 floats, 3000 samples, minibatches, the
    split-K slab path of gamma / beta: one PPO iteration against the oracle; a
    pickle round trip keeps the option."""
    import pickle

    from garage_amd._dtypes import EpisodeBatch, StepType
    from garage_amd.algos import PPO
    from garage_amd.optimizers import OptimizerWrapper
    from garage_amd.policies import GaussianMLPPolicy, GaussianMLPValueFunction
    from oracle import batch as ob
    from oracle.ppo import OraclePPO
    O, A, P = 17, 6, 32
    spec = _spec(O, A, P)
    torch.manual_seed(9)
    rng = np.random.RandomState(9)
    pol = GaussianMLPPolicy(spec, hidden_sizes=(256, 256),
                            layer_normalization=True)
    vf = GaussianMLPValueFunction(spec, hidden_sizes=(256, 256),
                                  layer_normalization=True)
    # gamma / beta away from their (1, 0) initialisation
    for net in (pol.net, vf.net):
        gen = torch.Generator(device='cpu').manual_seed(3)
        for name, view in net.named_views():
            if 'layer_normalization' in name:
                view.add_(0.2 * torch.randn(view.shape, generator=gen).to(
                    view.device))
    lens = rng.randint(4, P + 1, size=150)
    lens[::5] = P
    S = int(lens.sum())
    st = []
    for L in lens:
        t = [1] * L
        t[0] = 0
        t[-1] = 3 if L == P else 2
        st += t
    obs = (2.0 * rng.randn(S, O) + 0.5).astype(np.float32)
    acts = rng.randn(S, A).astype(np.float32)
    rew = rng.randn(S)
    E, mb = 2, 700
    oracle = OraclePPO(OrderedDict(pol.state_dict()), OrderedDict(vf.state_dict()),
                       max_episode_length=P, max_optimization_epochs=E,
                       minibatch_size=mb, policy_lr=1e-3, vf_lr=1e-3)
    b = ob.OracleEpisodeBatch(
        observations=obs, last_observations=np.zeros((len(lens), O), np.float32),
        actions=acts, rewards=rew, step_types=np.asarray(st), lengths=lens,
        max_episode_length=P)
    np.random.seed(4)
    want = oracle.train_once(b)
    wpol, wvf = oracle.state()
    pol2 = pickle.loads(pickle.dumps(pol))
    assert pol2.net.layer_norm and sorted(pol2.state_dict()) == sorted(
        pol.state_dict())
    algo = PPO(env_spec=spec, policy=pol, value_function=vf, sampler=None,
               policy_optimizer=OptimizerWrapper(
                   (torch.optim.Adam, dict(lr=1e-3)), pol,
                   max_optimization_epochs=E, minibatch_size=mb),
               vf_optimizer=OptimizerWrapper(
                   (torch.optim.Adam, dict(lr=1e-3)), vf,
                   max_optimization_epochs=E, minibatch_size=mb))
    batch = EpisodeBatch(env_spec=spec, episode_infos={}, observations=obs,
                         last_observations=np.zeros((len(lens), O), np.float32),
                         actions=acts, rewards=rew, env_infos={}, agent_infos={},
                         step_types=np.asarray([StepType(s) for s in st],
                                               dtype=object),
                         lengths=lens.astype('l'))
    np.random.seed(4)
    algo._train_once(0, batch)
    for k in ('policy/LossBefore', 'policy/LossAfter', 'policy/KL',
              'policy/Entropy', 'vf/LossBefore', 'vf/LossAfter'):
        assert np.isclose(algo.last_tabular[k], want[k], atol=2e-5,
                          rtol=2e-4), (k, algo.last_tabular[k], want[k])
    for mine, theirs in ((pol.state_dict(), wpol), (vf.state_dict(), wvf)):
        for k, v in mine.items():
            d = np.abs(v.numpy() - np.asarray(theirs[k]))
            assert d.max() <= 1e-4 and d.mean() <= 2e-6, (k, d.max(), d.mean())


@pytest.mark.parametrize('act', ['relu', 'none'])
def test_hidden_nonlinearity_at_c3_shape_against_oracle(act):
    """A relu / linear MLP(256,256) on a 3000-sample batch (the MFMA tile kernels,
    split-K slabs, the streaming weight-gradient kernels without their tanh'
    fusion): one PPO iteration against the oracle."""
    from garage_amd._dtypes import EpisodeBatch, StepType
    from garage_amd.algos import PPO
    from garage_amd.optimizers import OptimizerWrapper
    from garage_amd.policies import GaussianMLPPolicy, GaussianMLPValueFunction
    from oracle import batch as ob
    from oracle import networks as nets
    from oracle.ppo import OraclePPO
    fn = torch.relu if act == 'relu' else None
    O, A, P = 17, 6, 32
    spec = _spec(O, A, P)
    torch.manual_seed(8)
    rng = np.random.RandomState(8)
    pol = GaussianMLPPolicy(spec, hidden_sizes=(256, 256), hidden_nonlinearity=fn)
    vf = GaussianMLPValueFunction(spec, hidden_sizes=(256, 256),
                                  hidden_nonlinearity=fn)
    lens = rng.randint(4, P + 1, size=150)
    lens[::5] = P
    S = int(lens.sum())
    st = []
    for L in lens:
        t = [1] * L
        t[0] = 0
        t[-1] = 3 if L == P else 2
        st += t
    obs = rng.randn(S, O).astype(np.float32)
    acts = rng.randn(S, A).astype(np.float32)
    rew = rng.randn(S)
    E, mb = 2, 700
    with nets.hidden_nonlinearity(policy=fn, value=fn):
        oracle = OraclePPO(OrderedDict(pol.state_dict()),
                           OrderedDict(vf.state_dict()), max_episode_length=P,
                           max_optimization_epochs=E, minibatch_size=mb,
                           policy_lr=1e-3, vf_lr=1e-3)
        b = ob.OracleEpisodeBatch(
            observations=obs, last_observations=np.zeros((len(lens), O),
                                                         np.float32),
            actions=acts, rewards=rew, step_types=np.asarray(st), lengths=lens,
            max_episode_length=P)
        np.random.seed(4)
        want = oracle.train_once(b)
        wpol, wvf = oracle.state()
    algo = PPO(env_spec=spec, policy=pol, value_function=vf, sampler=None,
               policy_optimizer=OptimizerWrapper(
                   (torch.optim.Adam, dict(lr=1e-3)), pol,
                   max_optimization_epochs=E, minibatch_size=mb),
               vf_optimizer=OptimizerWrapper(
                   (torch.optim.Adam, dict(lr=1e-3)), vf,
                   max_optimization_epochs=E, minibatch_size=mb))
    batch = EpisodeBatch(env_spec=spec, episode_infos={}, observations=obs,
                         last_observations=np.zeros((len(lens), O), np.float32),
                         actions=acts, rewards=rew, env_infos={}, agent_infos={},
                         step_types=np.asarray([StepType(s) for s in st],
                                               dtype=object),
                         lengths=lens.astype('l'))
    np.random.seed(4)
    algo._train_once(0, batch)
    for k in ('policy/LossBefore', 'policy/LossAfter', 'policy/KL',
              'policy/Entropy', 'vf/LossBefore', 'vf/LossAfter'):
        assert np.isclose(algo.last_tabular[k], want[k], atol=2e-5,
                          rtol=2e-4), (k, algo.last_tabular[k], want[k])
    for mine, theirs in ((pol.state_dict(), wpol), (vf.state_dict(), wvf)):
        for k, v in mine.items():
            d = np.abs(v.numpy() - np.asarray(theirs[k]))
            assert d.max() <= 1e-4 and d.mean() <= 2e-6, (k, d.max(), d.mean())


SOFTPLUS_CASES = {
    'softplus': dict(std_parameterization='softplus', init_std=0.5),
    'softplus_max_clamp': dict(std_parameterization='softplus', max_std=0.8,
                               init_std=1.0),
}


@pytest.mark.parametrize('fused', [True, False])
@pytest.mark.parametrize('tag', sorted(SOFTPLUS_CASES))
def test_softplus_std_matches_real_reference(golden, tag, fused):
    """``std_parameterization='softplus'`` (``gaussian_mlp_module.py:178-181``:
    std = log(1 + exp(exp(p))), clamp on ``p`` first): the stored log-std and two
    real PPO iterations, through the one-launch small-minibatch step (default for
    these shapes) and through the per-layer loss kernels."""
    from garage_amd import _lib
    from garage_amd.algos import PPO
    from garage_amd.optimizers import OptimizerWrapper
    from garage_amd.policies import GaussianMLPPolicy, GaussianMLPValueFunction
    lib = _lib.load()
    g = golden('policy_options')
    O, A, P, E, mb = [int(v) for v in g[tag + '_cfg']]
    spec = _spec(O, A, P)
    pol = GaussianMLPPolicy(spec, hidden_sizes=(8, 8), **SOFTPLUS_CASES[tag])
    vf = GaussianMLPValueFunction(spec, hidden_sizes=(8, 8))
    pol.load_state_dict(_sd(g, tag + '_pol0:'))
    vf.load_state_dict(_sd(g, tag + '_vf0:'))
    x = torch.from_numpy(g[tag + '_fwd_obs'])
    dist, info = pol.forward(x)
    assert np.allclose(dist.mean.cpu().numpy(), g[tag + '_fwd_mean'], atol=2e-6)
    assert np.allclose(info['log_std'].cpu().numpy(), g[tag + '_fwd_log_std'],
                       atol=2e-6)
    algo = PPO(env_spec=spec, policy=pol, value_function=vf, sampler=None,
               policy_optimizer=OptimizerWrapper(
                   (torch.optim.Adam, dict(lr=2.5e-3)), pol,
                   max_optimization_epochs=E, minibatch_size=mb),
               vf_optimizer=OptimizerWrapper(
                   (torch.optim.Adam, dict(lr=2.5e-3)), vf,
                   max_optimization_epochs=E, minibatch_size=mb))
    try:
        lib.ga_set_small_step(1 if fused else 0)
        lib.ga_set_fused_train(1 if fused else 0)
        for it in range(2):
            pre = '%s_it%d_' % (tag, it)
            batch = _host_batch(spec, g, pre, O)
            np.random.seed(int(g[pre + 'np_seed']))
            algo._train_once(it, batch)
            for mine, theirs in LOG_KEYS.items():
                want = float(g[pre + 'log:' + theirs])
                assert np.isclose(algo.last_tabular[mine], want, atol=1e-5,
                                  rtol=1e-5), (mine, it, algo.last_tabular[mine],
                                               want)
            for k, v in pol.state_dict().items():
                assert np.allclose(v.numpy(), g[pre + 'pol:' + k], atol=2e-6), k
            for k, v in vf.state_dict().items():
                assert np.allclose(v.numpy(), g[pre + 'vf:' + k], atol=2e-6), k
    finally:
        lib.ga_set_small_step(1)
        lib.ga_set_fused_train(1)


@pytest.mark.parametrize('path', ['small_step', 'narrow_step', 'fused_train',
                                  'per_layer'])
def test_softplus_std_through_every_update_path_against_oracle(path):
    """The softplus std (and its chain factor on the log-std gradient) in each
    kernel family that evaluates the Gaussian loss: the one-launch small-minibatch
    step, the one-launch narrow step, the fused GEMM-epilogue kernels, the
    per-layer loss kernels -- one PPO iteration with an entropy term against the
    oracle."""
    from garage_amd import _lib
    from garage_amd._dtypes import EpisodeBatch, StepType
    from garage_amd.algos import PPO
    from garage_amd.optimizers import OptimizerWrapper
    from garage_amd.policies import GaussianMLPPolicy, GaussianMLPValueFunction
    from oracle import batch as ob
    from oracle import networks as nets
    from oracle.ppo import OraclePPO
    lib = _lib.load()
    hidden, mb = {'small_step': ((64, 64), 50), 'narrow_step': ((64, 64), 300),
                  'fused_train': ((128, 128), 500),
                  'per_layer': ((128, 128), 500)}[path]
    O, A, P = 11, 3, 24
    spec = _spec(O, A, P)
    torch.manual_seed(12)
    rng = np.random.RandomState(12)
    pol = GaussianMLPPolicy(spec, hidden_sizes=hidden, init_std=0.6,
                            std_parameterization='softplus')
    vf = GaussianMLPValueFunction(spec, hidden_sizes=hidden)
    lens = rng.randint(4, P + 1, size=70)
    lens[::5] = P
    S = int(lens.sum())
    st = []
    for L in lens:
        t = [1] * L
        t[0] = 0
        t[-1] = 3 if L == P else 2
        st += t
    obs = rng.randn(S, O).astype(np.float32)
    acts = rng.randn(S, A).astype(np.float32)
    rew = rng.randn(S)
    E = 2
    kw = dict(entropy_method='regularized', policy_ent_coeff=0.02)
    with nets.std_parameterization('softplus'):
        oracle = OraclePPO(OrderedDict(pol.state_dict()),
                           OrderedDict(vf.state_dict()), max_episode_length=P,
                           max_optimization_epochs=E, minibatch_size=mb,
                           policy_lr=1e-3, vf_lr=1e-3, **kw)
        b = ob.OracleEpisodeBatch(
            observations=obs, last_observations=np.zeros((len(lens), O),
                                                         np.float32),
            actions=acts, rewards=rew, step_types=np.asarray(st), lengths=lens,
            max_episode_length=P)
        np.random.seed(4)
        want = oracle.train_once(b)
        wpol, wvf = oracle.state()
    algo = PPO(env_spec=spec, policy=pol, value_function=vf, sampler=None,
               policy_optimizer=OptimizerWrapper(
                   (torch.optim.Adam, dict(lr=1e-3)), pol,
                   max_optimization_epochs=E, minibatch_size=mb),
               vf_optimizer=OptimizerWrapper(
                   (torch.optim.Adam, dict(lr=1e-3)), vf,
                   max_optimization_epochs=E, minibatch_size=mb), **kw)
    batch = EpisodeBatch(env_spec=spec, episode_infos={}, observations=obs,
                         last_observations=np.zeros((len(lens), O), np.float32),
                         actions=acts, rewards=rew, env_infos={}, agent_infos={},
                         step_types=np.asarray([StepType(s) for s in st],
                                               dtype=object),
                         lengths=lens.astype('l'))
    launches0 = lib.ga_small_step_launches()
    try:
        if path == 'per_layer':
            lib.ga_set_fused_train(0)
            lib.ga_set_small_step(0)
        np.random.seed(4)
        algo._train_once(0, batch)
    finally:
        lib.ga_set_fused_train(1)
        lib.ga_set_small_step(1)
    assert (lib.ga_small_step_launches() > launches0) == (path == 'small_step')
    for k in ('policy/LossBefore', 'policy/LossAfter', 'policy/KL',
              'policy/Entropy', 'vf/LossBefore', 'vf/LossAfter'):
        assert np.isclose(algo.last_tabular[k], want[k], atol=2e-5,
                          rtol=2e-4), (k, algo.last_tabular[k], want[k])
    for mine, theirs in ((pol.state_dict(), wpol), (vf.state_dict(), wvf)):
        for k, v in mine.items():
            d = np.abs(v.numpy() - np.asarray(theirs[k]))
            assert d.max() <= 1e-4 and d.mean() <= 2e-6, (k, d.max(), d.mean())
    # the log-std parameter moved, and by what the oracle says
    k = '_module._init_std'
    assert abs(float(pol.state_dict()[k]) - math.log(0.6)) > 1e-4
    assert np.isclose(float(pol.state_dict()[k]), float(np.asarray(wpol[k])),
                      atol=2e-6)


@pytest.mark.parametrize('hidden', [(32, 32), (300, )])
def test_rollout_samples_with_the_softplus_std(hidden):
    """Both sampling kernels (the fused rollout step for nets up to 256 wide, the
    head kernel behind the per-layer forward otherwise) draw actions with std =
    log(1 + exp(exp(p))): the spread of action - mean over ~7000 draws, and the
    stored ``agent_infos['log_std']``."""
    from garage_amd.envs import SyntheticVecEnv
    from garage_amd.policies import GaussianMLPPolicy
    from garage_amd.sampler import GpuVecSampler, GpuVecWorker
    n, O, A, P = 96, 5, 3, 24
    torch.manual_seed(3)
    env = SyntheticVecEnv(n, O, A, P, seed=9)
    pol = GaussianMLPPolicy(env.spec, hidden_sizes=hidden, init_std=0.4,
                            std_parameterization='softplus')
    want_std = math.log1p(math.exp(0.4))  # exp(p) = 0.4
    sampler = GpuVecSampler(agents=pol, envs=env, max_episode_length=P,
                            n_workers=1, worker_class=GpuVecWorker,
                            worker_args=dict(n_envs=n))
    eps = sampler.obtain_samples(0, n * P, agent_update=None)
    resid = np.asarray(eps.actions) - np.asarray(eps.agent_infos['mean'])
    assert resid.size >= 6000
    assert abs(resid.std() / want_std - 1.0) < 0.05, (resid.std(), want_std)
    assert np.allclose(np.asarray(eps.agent_infos['log_std']),
                       math.log(want_std), atol=1e-6)
    # (with the exp parameterisation the same parameter would give std 0.4)
    assert abs(resid.std() / 0.4 - 1.0) > 0.5


def test_rollout_with_relu_policy_stores_the_relu_means():
    """The fused rollout step implements tanh: a relu policy samples through the
    per-layer forward + head kernels; the stored ``agent_infos['mean']`` are the
    relu network's outputs on the stored observations (oracle forward)."""
    from garage_amd.envs import SyntheticVecEnv
    from garage_amd.policies import GaussianMLPPolicy
    from garage_amd.sampler import GpuVecSampler, GpuVecWorker
    from oracle import networks as nets
    n, O, A, P = 48, 7, 3, 12
    torch.manual_seed(2)
    env = SyntheticVecEnv(n, O, A, P, min_len=4, seed=5)
    pol = GaussianMLPPolicy(env.spec, hidden_sizes=(32, 32),
                            hidden_nonlinearity=torch.relu)
    sampler = GpuVecSampler(agents=pol, envs=env, max_episode_length=P,
                            n_workers=1, worker_class=GpuVecWorker,
                            worker_args=dict(n_envs=n))
    eps = sampler.obtain_samples(0, n * P, agent_update=None)
    params = OrderedDict(pol.state_dict())
    with nets.hidden_nonlinearity(policy=torch.relu), torch.no_grad():
        want = nets.policy_forward(
            params, torch.from_numpy(np.asarray(eps.observations)))[0].mean
    assert np.allclose(np.asarray(eps.agent_infos['mean']), want.numpy(),
                       atol=2e-6)
    # ... and they differ from what a tanh network with the same weights gives
    with torch.no_grad():
        tanh_mean = nets.policy_forward(
            params, torch.from_numpy(np.asarray(eps.observations)))[0].mean
    assert not np.allclose(want.numpy(), tanh_mean.numpy(), atol=1e-3)


SHAPE_CASES = [
    # (obs, act, policy hidden, value hidden, kwargs)
    (33, 17, (100, 37), (24, ), dict()),
    (3, 1, (5, ), (130, 7), dict(positive_adv=True)),
    (40, 2, (64, 64), (40, 40), dict(entropy_method='regularized',
                                     policy_ent_coeff=0.01)),
    (9, 30, (48, ), (33, 33, 33), dict()),
    (65, 4, (129, ), (64, ), dict(center_adv=False)),
]


@pytest.mark.parametrize('case', range(len(SHAPE_CASES)))
def test_train_once_matches_oracle_on_awkward_shapes(case):
    """Widths that exercise every dispatch edge at once (not multiples of 4 / 32
    / 64 / 128, action widths above the streaming kernels' limit, single hidden
    layers, deeper value nets): one PPO iteration with minibatches against the
    oracle, parameters and logged scalars."""
    from garage_amd._dtypes import EpisodeBatch, StepType
    from garage_amd.algos import PPO
    from garage_amd.optimizers import OptimizerWrapper
    from garage_amd.policies import GaussianMLPPolicy, GaussianMLPValueFunction
    from oracle import batch as ob
    from oracle.ppo import OraclePPO
    O, A, hp, hv, kw = SHAPE_CASES[case]
    P = 9
    spec = _spec(O, A, P)
    torch.manual_seed(case)
    rng = np.random.RandomState(case)
    pol = GaussianMLPPolicy(spec, hidden_sizes=hp)
    vf = GaussianMLPValueFunction(spec, hidden_sizes=hv)
    lens = rng.randint(1, P + 1, size=60)
    lens[0] = P
    S = int(lens.sum())
    st = []
    for L in lens:
        t = [1] * L
        t[0] = 0
        t[-1] = 3 if L == P else 2
        st += t
    obs = rng.randn(S, O).astype(np.float32)
    act = rng.randn(S, A).astype(np.float32)
    rew = rng.randn(S)
    E, mb = 2, 97
    oracle = OraclePPO(OrderedDict(pol.state_dict()),
                       OrderedDict(vf.state_dict()), max_episode_length=P,
                       max_optimization_epochs=E, minibatch_size=mb,
                       policy_lr=1e-3, vf_lr=1e-3, **kw)
    algo = PPO(env_spec=spec, policy=pol, value_function=vf, sampler=None,
               policy_optimizer=OptimizerWrapper(
                   (torch.optim.Adam, dict(lr=1e-3)), pol, E, mb),
               vf_optimizer=OptimizerWrapper(
                   (torch.optim.Adam, dict(lr=1e-3)), vf, E, mb), **kw)
    b = ob.OracleEpisodeBatch(observations=obs,
                              last_observations=np.zeros((len(lens), O),
                                                         np.float32),
                              actions=act, rewards=rew,
                              step_types=np.asarray(st), lengths=lens,
                              max_episode_length=P)
    np.random.seed(123)
    want = oracle.train_once(b)
    batch = EpisodeBatch(env_spec=spec, episode_infos={}, observations=obs,
                         last_observations=np.zeros((len(lens), O), np.float32),
                         actions=act, rewards=rew, env_infos={},
                         agent_infos={},
                         step_types=np.asarray([StepType(s) for s in st],
                                               dtype=object),
                         lengths=lens.astype('l'))
    np.random.seed(123)
    algo._train_once(0, batch)
    for k in LOG_KEYS:
        assert np.isclose(algo.last_tabular[k], want[k], atol=2e-5,
                          rtol=2e-5), (k, algo.last_tabular[k], want[k])
    wp, wv = oracle.state()
    for k, v in pol.state_dict().items():
        assert np.allclose(v.numpy(), wp[k], atol=3e-6), k
    for k, v in vf.state_dict().items():
        assert np.allclose(v.numpy(), wv[k], atol=3e-6), k
