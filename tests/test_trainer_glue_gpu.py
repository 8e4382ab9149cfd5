"""SURVEY.md row a29 / section 8b "Algorithm": garage's ``Trainer`` drives an
algorithm through ``algo._sampler``, ``algo.policy.get_param_values()``,
``sampler.obtain_samples(itr, batch_size, agent_update=..., env_update=...)``,
``algo.train(trainer)`` and a pickle of the algorithm after every epoch
(``trainer.py:153-160,179-229,263-341,361-455``).  ``MiniTrainer`` below restates
exactly those calls (test harness; ``Trainer`` itself is the caller's and is not
replaced) and runs PPO and TRPO through them, including a resume from the
snapshot of an earlier epoch."""
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class MiniTrainer:
    """The calls ``garage.Trainer`` makes on the algorithm and its sampler."""

    def __init__(self):
        self.total_env_steps = 0
        self.total_itr = 0
        self.snapshots = []
        self.step_itr = 0
        self.step_episode = None

    def setup(self, algo, env):                       # trainer.py:141-161
        self._algo, self._env = algo, env
        self._sampler = algo._sampler if hasattr(algo, '_sampler') else None

    def obtain_episodes(self, itr, batch_size=None, agent_update=None,
                        env_update=None):             # trainer.py:179-229
        if self._sampler is None:
            raise ValueError('trainer was not initialized with `sampler`.')
        if agent_update is None:
            policy = getattr(self._algo, 'exploration_policy', None)
            if policy is None:
                policy = self._algo.policy
            agent_update = policy.get_param_values()
        episodes = self._sampler.obtain_samples(
            itr, batch_size or self._batch_size, agent_update=agent_update,
            env_update=env_update)
        self.total_env_steps += sum(episodes.lengths)
        return episodes

    def step_epochs(self):                            # trainer.py:407-455
        self.step_itr = self.total_itr
        for epoch in range(self._start_epoch, self._n_epochs):
            yield epoch
            self.total_itr = self.step_itr
            # trainer.save -> snapshotter: the algorithm is pickled every epoch
            self.snapshots.append(pickle.dumps(
                dict(algo=self._algo, total_itr=self.total_itr,
                     total_env_steps=self.total_env_steps, epoch=epoch)))

    def train(self, n_epochs, batch_size, start_epoch=0):   # trainer.py:361-405
        self._n_epochs, self._batch_size = n_epochs, batch_size
        self._start_epoch = start_epoch
        average_return = self._algo.train(self)
        self._sampler.shutdown_worker()               # trainer.py:172-177
        return average_return


def _build(algo_name):
    from garage_amd.algos import PPO, TRPO
    from garage_amd.envs import SyntheticVecEnv
    from garage_amd.optimizers import OptimizerWrapper
    from garage_amd.policies import GaussianMLPPolicy, GaussianMLPValueFunction
    from garage_amd.sampler import GpuVecSampler, GpuVecWorker
    n, O, A, P = 32, 5, 2, 16
    torch.manual_seed(4)
    env = SyntheticVecEnv(n, O, A, P, min_len=5, seed=3)
    pol = GaussianMLPPolicy(env.spec, hidden_sizes=(32, 32))
    vf = GaussianMLPValueFunction(env.spec, hidden_sizes=(32, 32))
    sampler = GpuVecSampler(agents=pol, envs=env, max_episode_length=P,
                            n_workers=1, worker_class=GpuVecWorker,
                            worker_args=dict(n_envs=n))
    opt = (torch.optim.Adam, dict(lr=1e-3))
    kw = dict(env_spec=env.spec, policy=pol, value_function=vf,
              sampler=sampler,
              vf_optimizer=OptimizerWrapper(opt, vf, 2, 64,
                                            permutation='device', seed=5))
    if algo_name == 'ppo':
        algo = PPO(policy_optimizer=OptimizerWrapper(
            opt, pol, 2, 64, permutation='device', seed=6), **kw)
    else:
        algo = TRPO(**kw)
    return algo, env, n * P


@pytest.mark.parametrize('algo_name', ['ppo', 'trpo'])
def test_trainer_drives_train_and_resumes_from_a_snapshot(algo_name):
    algo, env, batch = _build(algo_name)
    trainer = MiniTrainer()
    trainer.setup(algo, env)
    assert trainer._sampler is algo._sampler
    p0 = algo.policy.net.params.clone()
    last = trainer.train(n_epochs=4, batch_size=batch)
    assert np.isfinite(last) and isinstance(float(last), float)
    assert trainer.total_itr == 4 and len(trainer.snapshots) == 4
    assert trainer.total_env_steps >= 4 * batch
    assert trainer.total_env_steps == algo._sampler.total_env_steps
    assert not torch.equal(algo.policy.net.params, p0)
    final_p = algo.policy.net.params.clone()
    final_v = algo._value_function.net.params.clone()

    # garage resume (trainer.py:263-341): restore the pickle taken after epoch 1
    # and run epochs 2..3 again -> the same bits as the uninterrupted run
    snap = pickle.loads(trainer.snapshots[1])
    algo2 = snap['algo']
    resumed = MiniTrainer()
    resumed.setup(algo2, env)
    resumed.total_itr = snap['total_itr']
    resumed.total_env_steps = snap['total_env_steps']
    resumed.train(n_epochs=4, batch_size=batch, start_epoch=snap['epoch'] + 1)
    assert resumed.total_itr == 4
    assert torch.equal(algo2.policy.net.params, final_p)
    assert torch.equal(algo2._value_function.net.params, final_v)
