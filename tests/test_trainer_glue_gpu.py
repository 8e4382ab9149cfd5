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
    """The calls ``garage.Trainer`` makes on the algorithm and its sampler, and
    its bookkeeping (``trainer.py:137-229,263-341,361-497``).  Its event log is
    held against ``tests/golden/trainer_trace.json``, recorded from the real
    ``Trainer`` (``test_mini_trainer_replays_the_real_trainers_trace``), so the
    other tests of this file drive the algorithms the way garage does."""

    class Stats:                                      # trainer.py:23-40

        def __init__(self):
            self.total_epoch = self.total_itr = self.total_env_steps = 0
            self.last_episode = None

    class TrainArgs:                                  # trainer.py:42-64

        def __init__(self, n_epochs, batch_size, start_epoch):
            self.n_epochs, self.batch_size = n_epochs, batch_size
            self.start_epoch = start_epoch
            self.store_episodes = False

    def __init__(self, snapshots=None, events=None):
        self._stats = MiniTrainer.Stats()
        self._train_args = None
        self._seed = None
        # (epoch, params) as handed to Snapshotter.save_itr_params; the algorithm
        # is pickled every epoch (snapshotter.py:85-123)
        self.snapshots = [] if snapshots is None else snapshots
        self.events = [] if events is None else events
        self.step_itr = None
        self.step_episode = None
        self._n_workers = self._worker_class = self._worker_args = None

    total_env_steps = property(lambda self: self._stats.total_env_steps)
    total_itr = property(lambda self: self._stats.total_itr)

    def setup(self, algo, env):                       # trainer.py:137-161
        self._algo, self._env = algo, env
        self._sampler = algo._sampler if hasattr(algo, '_sampler') else None
        self.events.append(['setup', self._sampler is getattr(
            algo, '_sampler', None)])

    def obtain_episodes(self, itr, batch_size=None, agent_update=None,
                        env_update=None):             # trainer.py:179-229
        if self._sampler is None:
            raise ValueError('trainer was not initialized with `sampler`.')
        if agent_update is None:
            policy = getattr(self._algo, 'exploration_policy', None)
            if policy is None:
                policy = self._algo.policy
            agent_update = policy.get_param_values()
        num = batch_size or self._train_args.batch_size
        current = self._algo.policy.state_dict()
        self.events.append([
            'obtain_samples', int(itr), int(num), sorted(agent_update.keys()),
            all(torch.equal(torch.as_tensor(v).cpu(), current[k].cpu())
                for k, v in agent_update.items()), env_update is None])
        episodes = self._sampler.obtain_samples(
            itr, num, agent_update=agent_update, env_update=env_update)
        self._stats.total_env_steps += sum(episodes.lengths)
        return episodes

    def save(self, epoch):                            # trainer.py:263-293
        params = dict(seed=self._seed, train_args=self._train_args,
                      stats=self._stats, env=self._env, algo=self._algo,
                      n_workers=self._n_workers,
                      worker_class=self._worker_class,
                      worker_args=self._worker_args)
        st, ta = self._stats, self._train_args
        self.events.append([
            'save', int(epoch), sorted(params.keys()), int(st.total_itr),
            int(st.total_env_steps), int(st.total_epoch),
            st.last_episode is None, params['algo'] is self._algo,
            int(ta.n_epochs), int(ta.batch_size), int(ta.start_epoch)])
        self.snapshots.append((epoch, pickle.dumps(dict(
            algo=self._algo, total_itr=st.total_itr,
            total_env_steps=int(st.total_env_steps), epoch=epoch,
            n_epochs=ta.n_epochs, batch_size=ta.batch_size))))

    def step_epochs(self):                            # trainer.py:407-455
        self.step_itr = self._stats.total_itr
        self.step_episode = None
        for epoch in range(self._train_args.start_epoch,
                           self._train_args.n_epochs):
            yield epoch
            self._stats.last_episode = (self.step_episode if
                                        self._train_args.store_episodes
                                        else None)
            self._stats.total_epoch = epoch
            self._stats.total_itr = self.step_itr
            self.save(epoch)

    def train(self, n_epochs, batch_size=None):       # trainer.py:361-405
        self._train_args = MiniTrainer.TrainArgs(n_epochs, batch_size, 0)
        average_return = self._algo.train(self)
        self._shutdown_worker()
        self.events.append(['train_returned', type(average_return).__name__,
                            int(self.total_env_steps), int(self.step_itr)])
        return average_return

    def _shutdown_worker(self):                       # trainer.py:172-177
        if self._sampler is not None:
            self.events.append(['shutdown_worker'])
            self._sampler.shutdown_worker()

    def restore(self, env, from_epoch='last'):        # trainer.py:295-341
        self.events.append(['load', from_epoch])
        epoch, blob = self.snapshots[-1] if from_epoch == 'last' else \
            next(s for s in self.snapshots if s[0] == from_epoch)
        saved = pickle.loads(blob)
        self._train_args = MiniTrainer.TrainArgs(saved['n_epochs'],
                                                 saved['batch_size'], 0)
        self._stats = MiniTrainer.Stats()
        self._stats.total_epoch = saved['epoch']
        self._stats.total_itr = saved['total_itr']
        self._stats.total_env_steps = saved['total_env_steps']
        self.setup(saved['algo'], env)
        self.events.pop()  # (the real restore logs nothing for its setup call)
        self._train_args.start_epoch = self._stats.total_epoch + 1
        self.events.append(['restored', int(self._train_args.start_epoch),
                            int(self._train_args.n_epochs),
                            int(self.total_env_steps)])
        return self._train_args

    def resume(self, n_epochs=None, batch_size=None):  # trainer.py:457-497
        self._train_args.n_epochs = n_epochs or self._train_args.n_epochs
        self._train_args.batch_size = (batch_size
                                       or self._train_args.batch_size)
        average_return = self._algo.train(self)
        self._shutdown_worker()
        self.events.append(['resume_returned', type(average_return).__name__,
                            int(self.total_env_steps), int(self.step_itr)])
        return average_return


def test_mini_trainer_replays_the_real_trainers_trace():
    """``tests/golden/trainer_trace.json``: the real ``garage.Trainer`` driving
    the real ``VPG`` (2 iterations per epoch) over 4 fixed-length envs for 2
    epochs, then restore + resume up to epoch 4 -- every call it made on the
    sampler (iteration numbers, batch size, the policy's ``get_param_values()``
    as agent update: same keys as the reference policy's ``state_dict``), what it
    put into each snapshot and its counters.  The same schedule through
    ``garage_amd.algos.VPG`` + ``GpuVecSampler`` under ``MiniTrainer`` must
    produce the same events."""
    import json
    import os

    from garage_amd.algos import VPG
    from garage_amd.envs import SyntheticVecEnv
    from garage_amd.policies import GaussianMLPPolicy, GaussianMLPValueFunction
    from garage_amd.sampler import GpuVecSampler, GpuVecWorker
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, 'golden', 'trainer_trace.json')) as f:
        ref = json.load(f)
    n, P = ref['n_envs'], ref['P']
    torch.manual_seed(0)
    env = SyntheticVecEnv(n, 3, 2, P, seed=1)  # fixed length P, as recorded
    pol = GaussianMLPPolicy(env.spec, hidden_sizes=tuple(ref['hidden']))
    vf = GaussianMLPValueFunction(env.spec, hidden_sizes=tuple(ref['hidden']))
    sampler = GpuVecSampler(agents=pol, envs=env, max_episode_length=P,
                            n_workers=1, worker_class=GpuVecWorker,
                            worker_args=dict(n_envs=n))
    algo = VPG(env_spec=env.spec, policy=pol, value_function=vf,
               sampler=sampler,
               num_train_per_epoch=ref['num_train_per_epoch'])
    trainer = MiniTrainer()
    trainer.setup(algo, env)
    trainer.train(n_epochs=2, batch_size=ref['batch_size'])
    resumed = MiniTrainer(snapshots=trainer.snapshots, events=trainer.events)
    resumed.restore(env)
    resumed.resume(n_epochs=4)
    got = trainer.events
    # (the real run saves the live algorithm object; after a restore from a pickle
    # `algo is algo` refers to the restored instance in both)
    assert len(got) == len(ref['events'])
    for mine, theirs in zip(got, ref['events']):
        assert mine == theirs, (mine, theirs)


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
    assert [e[0] for e in trainer.events if e[0] != 'obtain_samples'] == \
        ['setup', 'save', 'save', 'save', 'save', 'shutdown_worker',
         'train_returned']
    assert not torch.equal(algo.policy.net.params, p0)
    final_p = algo.policy.net.params.clone()
    final_v = algo._value_function.net.params.clone()

    # garage resume (trainer.py:263-341): restore the pickle taken after epoch 1
    # and run epochs 2..3 again -> the same bits as the uninterrupted run
    resumed = MiniTrainer(snapshots=trainer.snapshots)
    args = resumed.restore(env, from_epoch=1)
    assert args.start_epoch == 2 and args.n_epochs == 4
    algo2 = resumed._algo
    assert algo2 is not algo
    resumed.resume()
    assert resumed.total_itr == 4
    assert torch.equal(algo2.policy.net.params, final_p)
    assert torch.equal(algo2._value_function.net.params, final_v)
