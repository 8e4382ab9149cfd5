"""BASELINE.json configs[0] -- "CartPole-v1, garage.torch PPO, LocalSampler
n_workers=1, MLP(32,32), batch 2048" -- as a functional (plumbing) check: per-env
CPU ``Environment`` objects go through the batched-env adapter (``HostVecEnv``)
into the device sampler and PPO.  gym is not installed, so the cart-pole dynamics
below are this repository's own (the classic Barto-Sutton-Anderson equations,
Euler step 0.02 s); the check is the reference's own kind of assertion
(``tests/garage/torch/algos/test_ppo.py``: the run learns), not a benchmark.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class CartPole:
    """A garage-style ``Environment``: ``reset() -> (obs, info)``,
    ``step(a) -> EnvStep``-like with ``reward / observation / step_type``."""

    GRAVITY, M_CART, M_POLE, HALF_LEN, FORCE, TAU = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02
    X_LIMIT, THETA_LIMIT = 2.4, 12 * 2 * math.pi / 360

    def __init__(self, seed, max_episode_length=200):
        from garage_amd._dtypes import Box, Discrete, EnvSpec
        self._rng = np.random.RandomState(seed)
        self.spec = EnvSpec(Box(-np.inf, np.inf, (4, )), Discrete(2),
                            max_episode_length=max_episode_length)
        self._state = None
        self._t = 0

    def reset(self):
        self._state = self._rng.uniform(-0.05, 0.05, size=4)
        self._t = 0
        return self._state.astype(np.float32), {}

    def step(self, action):
        from garage_amd._dtypes import StepType
        from oracle.envs import EnvStepLite
        x, x_dot, th, th_dot = self._state
        force = self.FORCE if int(action) == 1 else -self.FORCE
        total = self.M_CART + self.M_POLE
        pml = self.M_POLE * self.HALF_LEN
        temp = (force + pml * th_dot**2 * math.sin(th)) / total
        th_acc = (self.GRAVITY * math.sin(th) - math.cos(th) * temp) / (
            self.HALF_LEN * (4.0 / 3.0 - self.M_POLE * math.cos(th)**2 / total))
        x_acc = temp - pml * th_acc * math.cos(th) / total
        self._state = np.array([x + self.TAU * x_dot, x_dot + self.TAU * x_acc,
                                th + self.TAU * th_dot,
                                th_dot + self.TAU * th_acc])
        self._t += 1
        done = (abs(self._state[0]) > self.X_LIMIT
                or abs(self._state[2]) > self.THETA_LIMIT)
        st = StepType.get_step_type(self._t, self.spec.max_episode_length, done)
        return EnvStepLite(action, 1.0, self._state.astype(np.float32), {}, st)

    def close(self):
        pass


@pytest.mark.timeout(600)
def test_cartpole_ppo_learns_through_the_host_env_adapter():
    from garage_amd.algos import PPO
    from garage_amd.optimizers import OptimizerWrapper
    from garage_amd.policies import (CategoricalMLPPolicy,
                                     GaussianMLPValueFunction)
    from garage_amd.sampler import GpuVecSampler, GpuVecWorker
    n_envs, batch = 16, 2048
    torch.manual_seed(0)
    np.random.seed(0)
    envs = [CartPole(seed=i) for i in range(n_envs)]
    spec = envs[0].spec
    pol = CategoricalMLPPolicy(spec, hidden_sizes=(32, 32))
    vf = GaussianMLPValueFunction(spec, hidden_sizes=(32, 32))
    sampler = GpuVecSampler(pol, [envs], max_episode_length=200, n_workers=1,
                            worker_class=GpuVecWorker, seed=1,
                            worker_args=dict(n_envs=n_envs))
    algo = PPO(env_spec=spec, policy=pol, value_function=vf, sampler=sampler,
               policy_optimizer=OptimizerWrapper(
                   (torch.optim.Adam, dict(lr=2.5e-3)), pol,
                   max_optimization_epochs=10, minibatch_size=64),
               vf_optimizer=OptimizerWrapper(
                   (torch.optim.Adam, dict(lr=2.5e-3)), vf,
                   max_optimization_epochs=10, minibatch_size=64),
               discount=0.99, gae_lambda=0.95, center_adv=True)
    returns = []
    for itr in range(12):
        eps = sampler.obtain_samples(itr, batch, None)
        lens = np.asarray(eps.lengths)
        assert int(lens.sum()) >= batch and lens.max() <= 200
        st = eps.step_types
        ends = np.cumsum(lens) - 1
        assert all(int(st[e]) in (2, 3) for e in ends)
        returns.append(float(algo._train_once(itr, eps)))
    assert np.isfinite(returns).all()
    # random play balances for ~20 steps; after 12 iterations of 2048 steps the
    # policy must be clearly better (the reference's tests assert return > 0)
    assert returns[0] < 40
    assert max(returns[-3:]) > 2.0 * returns[0], returns


def test_episode_infos_of_cpu_envs_reach_the_batch():
    """``reset()[1]`` of per-env CPU objects (goal-conditioned / multi-task envs)
    ends up in ``EpisodeBatch.episode_infos_by_episode`` as ``(N, ...)`` arrays,
    one row per episode in batch order -- ``DefaultWorker``'s layout
    (``sampler/default_worker.py:94-96,158-161``); the reference's ``VecWorker``
    loses them after an env's first episode."""
    from garage_amd._dtypes import Box, EnvSpec
    from garage_amd.policies import GaussianMLPPolicy
    from garage_amd.sampler import GpuVecSampler, GpuVecWorker
    from oracle.envs import CountingEnv
    P, n = 6, 5

    class GoalEnv(CountingEnv):

        def reset(self):
            obs, _ = super().reset()
            return obs, {'goal': np.array([self.env_id, self._episode],
                                          dtype=np.float32),
                         'task': self.env_id * 10 + self._episode}

    cyc = [[3, 6, 2], [4, 4, 4], [6, 1, 5], [2, 2, 6], [1, 3, 1]]
    envs = [GoalEnv(i, cyc[i], P) for i in range(n)]
    spec = EnvSpec(Box(-np.inf, np.inf, (3, )), Box(-np.inf, np.inf, (2, )),
                   max_episode_length=P)
    torch.manual_seed(0)
    pol = GaussianMLPPolicy(spec, hidden_sizes=(8, 8))
    for e in envs:
        e.spec = spec
    sampler = GpuVecSampler(pol, [envs], max_episode_length=P, n_workers=1,
                            worker_class=GpuVecWorker, seed=1,
                            worker_args=dict(n_envs=n))
    for call in range(2):  # the second call starts from in-flight episodes
        eps = sampler.obtain_samples(call, 60, None)
        lens = np.asarray(eps.lengths)
        starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
        first_obs = eps.observations[starts]          # [env, episode, t = 0]
        goal = eps.episode_infos_by_episode['goal']
        task = eps.episode_infos_by_episode['task']
        assert goal.shape == (len(lens), 2) and task.shape == (len(lens), )
        assert np.array_equal(goal, first_obs[:, :2])
        assert np.array_equal(task, first_obs[:, 0] * 10 + first_obs[:, 1])
        per_step = eps.episode_infos['goal']
        assert per_step.shape == (int(lens.sum()), 2)
        assert np.array_equal(per_step, eps.observations[:, :2])
        host = eps.to_host()
        assert np.array_equal(host.episode_infos_by_episode['goal'], goal)
