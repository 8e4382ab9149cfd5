"""The one-launch optimizer step for small minibatches (small_step.hip; the
reference's default minibatch is 64 samples, torch/algos/ppo.py:65-76): a PPO
iteration whose minibatch steps all take it, against the oracle (parameters and
logged scalars) and against the per-layer path of this library."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

LOG_KEYS = ('policy/LossBefore', 'policy/LossAfter', 'policy/KLBefore',
            'policy/KL', 'policy/Entropy', 'vf/LossBefore', 'vf/LossAfter')
CASES = {
    # O, A, hidden, minibatch, PPO keywords
    'c3_default_minibatch': (17, 6, (256, 256), 64, {}),
    'small_net': (4, 2, (64, 64), 64, {}),
    'ragged_rows': (9, 1, (128, 128), 48, {}),
    'vpg_objective': (5, 3, (64, 64), 33, {'vpg': True}),
    'uncentered': (17, 6, (192, 192), 64, {'center_adv': False}),
    'cartpole_sized': (4, 1, (32, 32), 64, {}),
    'entropy_regularized': (6, 3, (64, 64), 64,
                            {'entropy_method': 'regularized',
                             'policy_ent_coeff': 0.02}),
    'max_entropy_softplus': (6, 3, (64, 64), 40,
                             {'entropy_method': 'max', 'policy_ent_coeff': 0.01,
                              'center_adv': False, 'stop_entropy_gradient': True,
                              'use_softplus_entropy': True}),
    'odd_tiles': (12, 8, (96, 96), 50, {}),
}


def _problem(case):
    from garage_amd._dtypes import Box, EnvSpec, EpisodeBatch, StepType
    O, A, hidden, mb, kw = CASES[case]
    P = 9
    spec = EnvSpec(Box(-np.inf, np.inf, (O, )), Box(-np.inf, np.inf, (A, )),
                   max_episode_length=P)
    rng = np.random.RandomState(len(case))
    lens = rng.randint(1, P + 1, size=70)
    lens[0] = P
    S = int(lens.sum())
    st = []
    for L in lens:
        t = [1] * L
        t[0] = 0
        t[-1] = 3 if L == P else 2
        st += t
    data = dict(obs=rng.randn(S, O).astype(np.float32),
                act=rng.randn(S, A).astype(np.float32), rew=rng.randn(S),
                st=st, lens=lens)
    batch = EpisodeBatch(env_spec=spec, episode_infos={},
                         observations=data['obs'],
                         last_observations=np.zeros((len(lens), O), np.float32),
                         actions=data['act'], rewards=data['rew'], env_infos={},
                         agent_infos={},
                         step_types=np.asarray([StepType(s) for s in st],
                                               dtype=object),
                         lengths=lens.astype('l'))
    return spec, data, batch, P


def _algo(case, spec, seed=0):
    from garage_amd.algos import PPO, VPG
    from garage_amd.optimizers import OptimizerWrapper
    from garage_amd.policies import GaussianMLPPolicy, GaussianMLPValueFunction
    O, A, hidden, mb, kw = CASES[case]
    kw = dict(kw)
    cls = VPG if kw.pop('vpg', False) else PPO
    torch.manual_seed(seed)
    pol = GaussianMLPPolicy(spec, hidden_sizes=hidden)
    vf = GaussianMLPValueFunction(spec, hidden_sizes=hidden)
    opt = (torch.optim.Adam, dict(lr=1e-3))
    algo = cls(env_spec=spec, policy=pol, value_function=vf, sampler=None,
               policy_optimizer=OptimizerWrapper(opt, pol, 2, mb),
               vf_optimizer=OptimizerWrapper(opt, vf, 2, mb), **kw)
    return algo, pol, vf


@pytest.mark.parametrize('case', sorted(CASES))
def test_small_minibatch_steps_match_oracle_and_per_layer_path(case):
    from garage_amd import _lib
    from garage_amd.engine import reduction_workspace
    from oracle import batch as ob
    from oracle.ppo import OraclePPO
    lib = _lib.load()
    O, A, hidden, mb, kw = CASES[case]
    spec, data, batch, P = _problem(case)
    res = {}
    for on in (1, 0):
        lib.ga_set_small_step(on)
        algo, pol, vf = _algo(case, spec)
        if on:
            okw = {k: v for k, v in kw.items() if k != 'vpg'}
            if kw.get('vpg'):  # VPG: plain objective, gae_lambda defaults to 1
                okw.update(algo='vpg', gae_lambda=1.0)
            oracle = OraclePPO(OrderedDict(pol.state_dict()),
                               OrderedDict(vf.state_dict()),
                               max_episode_length=P, max_optimization_epochs=2,
                               minibatch_size=mb, policy_lr=1e-3, vf_lr=1e-3,
                               **okw)
        n0 = int(lib.ga_small_step_launches())
        np.random.seed(123)
        algo._train_once(0, batch)
        torch.cuda.synchronize()
        n_steps = 2 * 2 * -(-len(data['st']) // mb)  # 2 nets x 2 epochs
        assert int(lib.ga_small_step_launches()) - n0 == (n_steps if on else 0)
        for tag in (0, 1):
            assert reduction_workspace(pol.device, tag)[-2:].abs().sum() == 0
        res[on] = (pol.net.params.clone(), vf.net.params.clone(),
                   dict(algo.last_tabular), pol, vf)
    lib.ga_set_small_step(1)
    b = ob.OracleEpisodeBatch(
        observations=data['obs'],
        last_observations=np.zeros((len(data['lens']), O), np.float32),
        actions=data['act'], rewards=data['rew'],
        step_types=np.asarray(data['st']), lengths=data['lens'],
        max_episode_length=P)
    np.random.seed(123)
    want = oracle.train_once(b)
    wp, wv = oracle.state()
    _, _, tab, pol, vf = res[1]
    for k in LOG_KEYS:
        assert np.isclose(tab[k], want[k], atol=2e-5, rtol=2e-5), \
            (k, tab[k], want[k])
    # Adam divides by sqrt(v) + 1e-8: where a gradient is itself ~1e-8, its last
    # bits (another summation order) move the step by a sizeable part of lr, so
    # after 4 steps of lr = 1e-3 single elements may sit 1e-4 apart while the bulk
    # agrees to 1e-7.  The gradients themselves are compared tightly below.
    for state, ref in ((pol.state_dict(), wp), (vf.state_dict(), wv)):
        for k, v in state.items():
            d = np.abs(v.numpy() - np.asarray(ref[k]))
            assert d.max() < 5e-4, (k, d.max())
            assert d.mean() < 2e-7, (k, d.mean())
    for i in (0, 1):  # and the per-layer path of this library
        d = (res[1][i] - res[0][i]).abs()
        assert float(d.max()) < 5e-4 and float(d.mean()) < 2e-7


@pytest.mark.parametrize('case', sorted(CASES))
def test_small_step_gradients_match_per_layer_path(case):
    """Adam with beta1 = beta2 = 0 and eps = 1 moves a parameter by
    -lr g / (|g| + 1): linear in g for small gradients, so (new - old) / lr
    exposes the gradients of one pass (several steps, the last one ragged) of
    both paths to the resolution of an fp32 parameter."""
    from garage_amd import _lib
    from garage_amd.algos import PPO, VPG
    from garage_amd.optimizers import OptimizerWrapper
    from garage_amd.policies import GaussianMLPPolicy, GaussianMLPValueFunction
    lib = _lib.load()
    O, A, hidden, mb, kw = CASES[case]
    kw = dict(kw)
    cls = VPG if kw.pop('vpg', False) else PPO
    spec, data, batch, P = _problem(case)
    lr = 1e-3
    out = {}
    for on in (1, 0):
        lib.ga_set_small_step(on)
        torch.manual_seed(0)
        pol = GaussianMLPPolicy(spec, hidden_sizes=hidden)
        vf = GaussianMLPValueFunction(spec, hidden_sizes=hidden)
        opt = (torch.optim.Adam, dict(lr=lr, betas=(0.0, 0.0), eps=1.0))
        algo = cls(env_spec=spec, policy=pol, value_function=vf, sampler=None,
                   policy_optimizer=OptimizerWrapper(opt, pol, 1, mb),
                   vf_optimizer=OptimizerWrapper(opt, vf, 1, mb), **kw)
        p0, v0 = pol.net.params.clone(), vf.net.params.clone()
        np.random.seed(123)
        algo._train_once(0, batch)
        out[on] = ((pol.net.params - p0) / lr, (vf.net.params - v0) / lr)
    lib.ga_set_small_step(1)
    for i in (0, 1):
        scale = float(out[0][i].abs().max())
        assert scale > 1e-2  # the pass did move the parameters
        d = float((out[1][i] - out[0][i]).abs().max())
        assert d < 1e-4 * scale + 4e-5, (i, d, scale)


@pytest.mark.parametrize('kw', [dict(), dict(entropy_method='regularized',
                                             policy_ent_coeff=0.02),
                                dict(entropy_method='max', policy_ent_coeff=0.05,
                                     center_adv=False,
                                     stop_entropy_gradient=True)])
def test_categorical_small_steps_vs_oracle_and_per_layer_path(kw):
    """BASELINE.json configs[0]'s shape of problem (discrete actions, MLP(32,32),
    the default minibatch of 64): a categorical PPO iteration whose optimizer
    steps all take the one-launch path, against the oracle and -- gradients, with
    the linear-regime Adam -- against the per-layer path."""
    from garage_amd import _lib
    from garage_amd.algos import PPO
    from garage_amd.envs import SyntheticVecEnv
    from garage_amd.optimizers import OptimizerWrapper
    from garage_amd.policies import (CategoricalMLPPolicy,
                                     GaussianMLPValueFunction)
    from garage_amd.sampler import GpuVecSampler, GpuVecWorker
    from oracle import batch as ob
    from oracle.ppo import OraclePPO
    lib = _lib.load()
    n, O, A, P = 24, 4, 2, 20
    E, mb = 2, 64

    def build(opt):
        torch.manual_seed(9)
        env = SyntheticVecEnv(n, O, A, P, min_len=4, seed=11, discrete=True)
        pol = CategoricalMLPPolicy(env.spec, hidden_sizes=(32, 32))
        vf = GaussianMLPValueFunction(env.spec, hidden_sizes=(32, 32))
        sampler = GpuVecSampler(pol, env, max_episode_length=P, n_workers=1,
                                worker_class=GpuVecWorker,
                                worker_args=dict(n_envs=n))
        algo = PPO(env_spec=env.spec, policy=pol, value_function=vf,
                   sampler=sampler,
                   policy_optimizer=OptimizerWrapper(opt, pol, E, mb),
                   vf_optimizer=OptimizerWrapper(opt, vf, E, mb), **kw)
        return algo, sampler, pol, vf

    # against the oracle, default Adam
    lib.ga_set_small_step(1)
    algo, sampler, pol, vf = build((torch.optim.Adam, dict(lr=1e-3)))
    oracle = OraclePPO(pol.state_dict(), vf.state_dict(), max_episode_length=P,
                       policy_kind='categorical', max_optimization_epochs=E,
                       minibatch_size=mb, policy_lr=1e-3, vf_lr=1e-3, **kw)
    eps = sampler.obtain_samples(0, n * P, None)
    host = ob.OracleEpisodeBatch(
        observations=eps.observations, last_observations=eps.last_observations,
        actions=eps.actions, rewards=eps.rewards, step_types=eps.step_types,
        lengths=eps.lengths, max_episode_length=P)
    np.random.seed(70)
    want = oracle.train_once(host)
    n0 = int(lib.ga_small_step_launches())
    np.random.seed(70)
    algo._train_once(0, eps)
    S = int(eps.lengths.sum())
    assert int(lib.ga_small_step_launches()) - n0 == 2 * E * -(-S // mb)
    for k in LOG_KEYS:
        assert np.isclose(algo.last_tabular[k], want[k], atol=2e-5, rtol=2e-5), \
            (k, algo.last_tabular[k], want[k])
    wp, wv = oracle.state()
    for state, ref in ((pol.state_dict(), wp), (vf.state_dict(), wv)):
        for k, v in state.items():
            d = np.abs(v.numpy() - np.asarray(ref[k]))
            assert d.max() < 5e-4 and d.mean() < 1e-6, (k, d.max(), d.mean())
    # gradients against the per-layer path
    out = {}
    for on in (1, 0):
        lib.ga_set_small_step(on)
        algo, sampler, pol, vf = build(
            (torch.optim.Adam, dict(lr=1e-3, betas=(0.0, 0.0), eps=1.0)))
        eps = sampler.obtain_samples(0, n * P, None)
        p0, v0 = pol.net.params.clone(), vf.net.params.clone()
        np.random.seed(70)
        algo._train_once(0, eps)
        out[on] = ((pol.net.params - p0) / 1e-3, (vf.net.params - v0) / 1e-3)
    lib.ga_set_small_step(1)
    for i in (0, 1):
        scale = float(out[0][i].abs().max())
        d = float((out[1][i] - out[0][i]).abs().max())
        assert scale > 1e-2 and d < 1e-4 * scale + 4e-5, (i, d, scale)


def test_forced_fallback_takes_the_per_layer_path():
    """Shapes whose two concurrent grids the device could not hold at once (the
    grid barriers need every workgroup resident) take the per-layer launches:
    forced here by telling the library the device holds no workgroup at all."""
    from garage_amd import _lib
    lib = _lib.load()
    case = 'c3_default_minibatch'
    spec, data, batch, P = _problem(case)
    res = {}
    try:
        for cap in (-1, 0):
            lib.ga_set_small_step_resident_cap(cap)
            algo, pol, vf = _algo(case, spec)
            n0 = int(lib.ga_small_step_launches())
            np.random.seed(5)
            algo._train_once(0, batch)
            torch.cuda.synchronize()
            launches = int(lib.ga_small_step_launches()) - n0
            assert (launches > 0) == (cap < 0)
            res[cap] = (pol.net.params.clone(), vf.net.params.clone())
    finally:
        lib.ga_set_small_step_resident_cap(-1)
    lib.ga_set_small_step(0)
    try:
        algo, pol, vf = _algo(case, spec)
        np.random.seed(5)
        algo._train_once(0, batch)
    finally:
        lib.ga_set_small_step(1)
    # the fallback IS the per-layer path: same bits
    assert torch.equal(res[0][0], pol.net.params)
    assert torch.equal(res[0][1], vf.net.params)
    for i in (0, 1):
        d = (res[-1][i] - res[0][i]).abs()
        assert float(d.max()) < 5e-4 and float(d.mean()) < 2e-7


def test_a_barrier_that_gives_up_leaves_the_parameters_untouched():
    """``max_polls = 0``: the first workgroup to wait at a grid barrier abandons
    the launch (one compare-and-swap decides for the whole grid).  Nothing may be
    written to parameters or moments by that launch or by the later ones of the
    iteration, ``_train_once`` must raise, and the next iteration must work."""
    from garage_amd import _lib
    lib = _lib.load()
    case = 'c3_default_minibatch'
    spec, data, batch, P = _problem(case)
    algo, pol, vf = _algo(case, spec)
    before = [t.clone() for net in (pol.net, vf.net)
              for t in (net.params, net.exp_avg, net.exp_avg_sq)]
    lib.ga_set_small_step_max_polls(0)
    try:
        np.random.seed(5)
        with pytest.raises(RuntimeError, match='grid barrier timed out'):
            algo._train_once(0, batch)
    finally:
        lib.ga_set_small_step_max_polls(-1)
    after = [t for net in (pol.net, vf.net)
             for t in (net.params, net.exp_avg, net.exp_avg_sq)]
    for a, b in zip(before, after):
        assert torch.equal(a, b)
    # re-armed: a fresh pair of networks trains as usual afterwards
    test_small_minibatch_steps_match_oracle_and_per_layer_path(case)
