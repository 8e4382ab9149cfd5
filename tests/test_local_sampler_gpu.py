"""tests/garage/sampler/test_local_sampler.py mirrored for ``GpuVecSampler``:
``obtain_exact_episodes`` with one agent per worker (worker order, episodes per
worker), a factory without a seed, and construction without a factory."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

P = 5


def _spec():
    from garage_amd._dtypes import Box, EnvSpec
    return EnvSpec(Box(-np.inf, np.inf, (3, )), Box(-np.inf, np.inf, (2, )),
                   max_episode_length=P)


def _env():
    from oracle import envs as oenvs
    env = oenvs.CountingEnv(0, [3, 5, 2], P)
    env.spec = _spec()
    return env


def _fixed_policy(action):
    """Deterministic 'policy': zero weights, bias = action, sigma -> 0."""
    from garage_amd.policies import GaussianMLPPolicy
    pol = GaussianMLPPolicy(_spec(), hidden_sizes=(), init_std=1e-6,
                            min_std=None)
    pol.net.weight(0).zero_()
    pol.net.bias(0).copy_(torch.tensor(action, dtype=torch.float32))
    return pol


def test_obtain_exact_episodes():
    """test_local_sampler.py:64-88: every worker contributes exactly
    ``n_eps_per_worker`` episodes, in worker order, acting with its own agent."""
    from garage_amd.sampler import GpuVecSampler, GpuVecWorker, WorkerFactory
    n_workers, n_eps = 4, 3
    rng = np.random.RandomState(0)
    actions = [rng.uniform(-1, 1, size=2).astype(np.float32)
               for _ in range(n_workers)]
    policies = [_fixed_policy(a) for a in actions]
    wf = WorkerFactory(seed=100, max_episode_length=P, n_workers=n_workers,
                       worker_class=GpuVecWorker,
                       worker_args=dict(n_envs=1))
    sampler = GpuVecSampler.from_worker_factory(wf, policies, envs=_env())
    eps = sampler.obtain_exact_episodes(n_eps, agent_update=policies)
    assert sum(eps.lengths) >= n_workers * n_eps
    assert len(eps.lengths) == n_workers * n_eps
    # a single environment is deep-copied per worker: every worker sees the
    # same cycle of episode lengths from its start
    assert np.array_equal(eps.lengths, [3, 5, 2] * n_workers)
    worker = -1
    for count, ep in enumerate(eps.split()):
        if count % n_eps == 0:
            worker += 1
        assert np.allclose(ep.actions, actions[worker], atol=1e-4), count
    assert sampler.total_env_steps == int(sum(eps.lengths))
    sampler.shutdown_worker()


def test_no_seed_and_init_without_worker_factory():
    """test_local_sampler.py:91-126."""
    from garage_amd.sampler import GpuVecSampler, GpuVecWorker, WorkerFactory
    pol = _fixed_policy([0.1, -0.2])
    wf = WorkerFactory(seed=None, max_episode_length=P, n_workers=2,
                       worker_class=GpuVecWorker, worker_args=dict(n_envs=2))
    sampler = GpuVecSampler.from_worker_factory(wf, pol, _env())
    eps = sampler.obtain_samples(0, 40, pol)
    assert sum(eps.lengths) >= 40
    sampler.shutdown_worker()

    sampler = GpuVecSampler(agents=pol, envs=_env(), seed=100,
                            max_episode_length=P)
    other = WorkerFactory(seed=100, max_episode_length=P)
    assert sampler._factory._seed == other._seed
    assert sampler._factory._max_episode_length == other._max_episode_length
    with pytest.raises(TypeError, match='Must construct a sampler from'):
        GpuVecSampler(agents=pol, envs=_env())
    eps = sampler.obtain_samples(0, 20, None)
    assert sum(eps.lengths) >= 20


def test_ownership_policy_by_reference_envs_by_copy_batches_not_reused():
    """``local_sampler.py:16-18,77-79``: the sampler works on the caller's policy
    object (updates of its parameters are seen without an ``agent_update``), on
    deep copies of the environments, and never touches a batch it has returned."""
    from garage_amd.sampler import GpuVecSampler, GpuVecWorker
    pol = _fixed_policy([0.25, -0.5])
    env = _env()
    sampler = GpuVecSampler(agents=pol, envs=env, max_episode_length=P,
                            n_workers=1, worker_class=GpuVecWorker,
                            worker_args=dict(n_envs=3))
    worker = sampler._workers[0]
    assert worker.agent is pol
    assert all(e is not env for e in worker.env.envs)
    assert len({id(e) for e in worker.env.envs}) == 3
    eps1 = sampler.obtain_samples(0, 20, None)
    kept = {k: np.array(getattr(eps1, k), copy=True)
            for k in ('observations', 'actions', 'rewards', 'lengths')}
    assert np.allclose(eps1.actions, [0.25, -0.5], atol=1e-4)
    assert env._episode == -1  # the caller's environment was never stepped
    # the caller changes its policy in place: the next batch follows
    pol.net.bias(0).copy_(torch.tensor([0.75, 0.1]))
    eps2 = sampler.obtain_samples(1, 20, None)
    assert np.allclose(eps2.actions, [0.75, 0.1], atol=1e-4)
    for k, v in kept.items():
        assert np.array_equal(getattr(eps1, k), v), k
    sampler.shutdown_worker()
