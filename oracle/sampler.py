"""Oracle: the reference's in-process samplers and workers, restated.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Restates the control flow
and bookkeeping of
  * ``sampler/default_worker.py:12-190``  DefaultWorker (one env, one episode
    per rollout; ``lengths`` dtype 'i'; TERMINAL keeps the *previous*
    observation as last_observation, Q11)
  * ``sampler/vec_worker.py:12-228``      VecWorker (n envs in lock step;
    rollout returns as soon as >=1 episode finished; per-call reset drops
    in-flight episodes, Q12; completion order = (step, env index), Q13;
    ``lengths`` dtype 'l')
  * ``sampler/fragment_worker.py:11-156`` + ``sampler/_dtypes.py:9-108``
    FragmentWorker / InProgressEpisode
  * ``sampler/local_sampler.py:134-205``  LocalSampler.obtain_samples /
    obtain_exact_episodes
  * ``sampler/worker_factory.py:68-116``  message canonicalisation.
It is also the rollout half of ``bench.py``'s ``cpu_baseline``.

``alias_bug`` (VecWorker only): the reference stores ``self._prev_obs[i]`` -- a
*view* of row i of a 2-D array it later overwrites in place -- so every stored
observation of an episode ends up equal to the episode's final observation
(SURVEY.md Q10).  ``alias_bug=True`` reproduces that (used to prove the oracle
matches the real VecWorker field by field); the default ``False`` stores what
DefaultWorker stores, which is what the product implements.
"""
import copy

import numpy as np

from oracle.batch import OracleEpisodeBatch, StepType


def _stack_infos(infos):
    return {k: np.asarray(v) for k, v in infos.items()}


class OracleDefaultWorker:
    """``default_worker.py``: one env, one episode per ``rollout()``."""

    def __init__(self, *, seed, max_episode_length, worker_number):
        self._seed = seed
        self._max_episode_length = max_episode_length
        self._worker_number = worker_number
        self.agent = None
        self.env = None

    def update_agent(self, agent_update):
        if agent_update is not None:
            self.agent = agent_update

    def update_env(self, env_update):
        if env_update is not None:
            self.env = env_update

    def rollout(self):
        obs_list, actions, rewards, step_types = [], [], [], []
        infos = {}
        prev_obs, _ = self.env.reset()
        self.agent.reset()
        length = 0
        while length < self._max_episode_length:
            a, info = self.agent.get_action(prev_obs)
            es = self.env.step(a)
            obs_list.append(prev_obs)
            actions.append(es.action)
            rewards.append(es.reward)
            step_types.append(es.step_type)
            for k, v in info.items():
                infos.setdefault(k, []).append(v)
            length += 1
            if es.terminal:
                break  # default_worker.py:117-121: _prev_obs is NOT advanced
            prev_obs = es.observation  # a TIMEOUT step does advance it
        return OracleEpisodeBatch(
            observations=np.asarray(obs_list),
            last_observations=np.asarray([prev_obs]),
            actions=np.asarray(actions),
            rewards=np.asarray(rewards),
            step_types=np.asarray(step_types, dtype=object),
            lengths=np.asarray([length], dtype='i'),
            agent_infos=_stack_infos(infos),
            max_episode_length=self._max_episode_length)

    def shutdown(self):
        self.env.close()


class OracleVecWorker:
    """``vec_worker.py``: ``n_envs`` copies stepped in lock step."""

    def __init__(self, *, seed, max_episode_length, worker_number, n_envs=8,
                 alias_bug=False):
        self._seed = seed
        self._max_episode_length = max_episode_length
        self._worker_number = worker_number
        self._n_envs = n_envs
        self._alias_bug = alias_bug
        self.agent = None
        self._envs = [None] * n_envs
        self._needs_agent_reset = True
        self._needs_env_reset = True
        self._completed = []
        self._lengths = [0] * n_envs

    def update_agent(self, agent_update):
        if agent_update is not None:
            self.agent = agent_update
        self._needs_agent_reset = True  # vec_worker.py:72-73

    def update_env(self, env_update):
        if isinstance(env_update, list):
            if len(env_update) != self._n_envs:
                raise ValueError('If separate environments are passed for '
                                 'each worker, there must be exactly n_envs '
                                 '({}) environments, but received {} '
                                 'environments.'.format(
                                     self._n_envs, len(env_update)))
        elif env_update is not None:
            env_update = [copy.deepcopy(env_update)
                          for _ in range(self._n_envs)]
        if env_update:
            for i, env in enumerate(env_update):
                self._envs[i] = env
            self._needs_env_reset = True

    def _clear(self, i):
        self._obs[i], self._act[i], self._rew[i], self._st[i] = [], [], [], []
        self._infos[i] = {}
        self._env_infos[i] = {}
        self._lengths[i] = 0

    def start_episode(self):
        """``vec_worker.py:107-137``."""
        if not (self._needs_agent_reset or self._needs_env_reset):
            return
        n = self._n_envs
        self.agent.reset([True] * n)
        if self._needs_env_reset:
            self._prev_obs = np.asarray([env.reset()[0] for env in self._envs])
        else:
            for i, env in enumerate(self._envs):
                if self._lengths[i] > 0:  # only envs with progress are reset
                    self._prev_obs[i] = env.reset()[0]
        self._obs = [[] for _ in range(n)]
        self._act = [[] for _ in range(n)]
        self._rew = [[] for _ in range(n)]
        self._st = [[] for _ in range(n)]
        self._infos = [{} for _ in range(n)]
        self._env_infos = [{} for _ in range(n)]
        self._lengths = [0] * n
        self._needs_agent_reset = False
        self._needs_env_reset = False

    def _gather(self, i, last_observation):
        """``vec_worker.py:139-174``."""
        self._completed.append(
            OracleEpisodeBatch(
                observations=np.asarray(self._obs[i]),
                last_observations=np.asarray([last_observation]),
                actions=np.asarray(self._act[i]),
                rewards=np.asarray(self._rew[i]),
                step_types=np.asarray(self._st[i], dtype=object),
                lengths=np.asarray([self._lengths[i]], dtype='l'),
                agent_infos=_stack_infos(self._infos[i]),
                env_infos=_stack_infos(self._env_infos[i]),
                max_episode_length=self._max_episode_length))
        self._clear(i)
        self._prev_obs[i] = self._envs[i].reset()[0]

    def step_episode(self):
        """``vec_worker.py:176-204``."""
        finished = False
        actions, agent_info = self.agent.get_actions(self._prev_obs)
        completes = [False] * self._n_envs
        for i, action in enumerate(actions):
            if self._lengths[i] < self._max_episode_length:
                es = self._envs[i].step(action)
                stored = self._prev_obs[i]
                if not self._alias_bug:
                    stored = np.array(stored, copy=True)
                self._obs[i].append(stored)
                self._rew[i].append(es.reward)
                self._act[i].append(es.action)
                for k, v in agent_info.items():
                    self._infos[i].setdefault(k, []).append(v[i])
                for k, v in es.env_info.items():  # vec_worker.py:192-193
                    self._env_infos[i].setdefault(k, []).append(v)
                self._lengths[i] += 1
                self._st[i].append(es.step_type)
                self._prev_obs[i] = es.observation
            if self._lengths[i] >= self._max_episode_length or es.last:
                self._gather(i, es.observation)
                completes[i] = True
                finished = True
        if finished:
            self.agent.reset(completes)
        return finished

    def collect_episode(self):
        done, self._completed = self._completed, []
        return done[0] if len(done) == 1 else \
            OracleEpisodeBatch.concatenate(*done)

    def rollout(self):
        self.start_episode()
        while not self.step_episode():
            pass
        return self.collect_episode()

    def shutdown(self):
        for env in self._envs:
            env.close()


class _Fragment:
    """``sampler/_dtypes.py:9-108`` InProgressEpisode."""

    def __init__(self, env, initial_observation=None):
        self.env = env
        if initial_observation is None:
            initial_observation, _ = env.reset()
        self.observations = [initial_observation]
        self.actions, self.rewards, self.step_types = [], [], []
        self.infos = {}

    @property
    def last_obs(self):
        return self.observations[-1]

    def step(self, action, agent_info):
        es = self.env.step(action)
        self.observations.append(es.observation)
        self.rewards.append(es.reward)
        self.actions.append(es.action)
        self.step_types.append(es.step_type)
        for k, v in agent_info.items():
            self.infos.setdefault(k, []).append(v)
        return es.observation

    def to_batch(self, max_episode_length):
        assert len(self.rewards) > 0
        return OracleEpisodeBatch(
            observations=np.asarray(self.observations[:-1]),
            last_observations=np.asarray([self.last_obs]),
            actions=np.asarray(self.actions),
            rewards=np.asarray(self.rewards),
            step_types=np.asarray(self.step_types, dtype=object),
            lengths=np.asarray([len(self.rewards)], dtype='l'),
            agent_infos=_stack_infos(self.infos),
            max_episode_length=max_episode_length)


class OracleFragmentWorker:
    """``fragment_worker.py``: ``timesteps_per_call`` steps of every env."""

    def __init__(self, *, seed, max_episode_length, worker_number, n_envs=8,
                 timesteps_per_call=1):
        self._max_episode_length = max_episode_length
        self._n_envs = n_envs
        self._timesteps_per_call = timesteps_per_call
        self._needs_env_reset = True
        self._envs = [None] * n_envs
        self._lengths = [0] * n_envs
        self._complete = []
        self._fragments = None
        self.agent = None

    def update_agent(self, agent_update):
        if agent_update is not None:
            self.agent = agent_update

    def update_env(self, env_update):
        if isinstance(env_update, list):
            if len(env_update) != self._n_envs:
                raise ValueError('wrong number of environments')
        elif env_update is not None:
            env_update = [copy.deepcopy(env_update)
                          for _ in range(self._n_envs)]
        if env_update:
            for i, env in enumerate(env_update):
                self._envs[i] = env
            self._needs_env_reset = True

    def start_episode(self):
        if self._needs_env_reset:
            self._needs_env_reset = False
            self.agent.reset([True] * self._n_envs)
            self._lengths = [0] * self._n_envs
            self._fragments = [_Fragment(env) for env in self._envs]

    def step_episode(self):
        prev = np.asarray([f.last_obs for f in self._fragments])
        actions, infos = self.agent.get_actions(prev)
        completes = [False] * self._n_envs
        for i, action in enumerate(actions):
            frag = self._fragments[i]
            if self._lengths[i] < self._max_episode_length:
                frag.step(action, {k: v[i] for k, v in infos.items()})
                self._lengths[i] += 1
            if (self._lengths[i] >= self._max_episode_length
                    or frag.step_types[-1] == StepType.TERMINAL):
                self._lengths[i] = 0
                self._complete.append(frag.to_batch(self._max_episode_length))
                self._fragments[i] = _Fragment(self._envs[i])
                completes[i] = True
        if any(completes):
            self.agent.reset(completes)
        return any(completes)

    def collect_episode(self):
        for i, frag in enumerate(self._fragments):
            if len(frag.rewards) > 0:
                self._complete.append(frag.to_batch(self._max_episode_length))
                self._fragments[i] = _Fragment(frag.env, frag.last_obs)
        assert self._complete
        out = OracleEpisodeBatch.concatenate(*self._complete)
        self._complete = []
        return out

    def rollout(self):
        self.start_episode()
        for _ in range(self._timesteps_per_call):
            self.step_episode()
        return self.collect_episode()

    def shutdown(self):
        for env in self._envs:
            env.close()


class OracleLocalSampler:
    """``local_sampler.py``: round-robin ``worker.rollout()`` in-process."""

    def __init__(self, agents, envs, *, max_episode_length, n_workers=1,
                 worker_class=OracleDefaultWorker, worker_args=None, seed=None):
        if max_episode_length is None:
            raise TypeError('Must construct a sampler from WorkerFactory or'
                            'parameters (at least max_episode_length)')
        self.n_workers = n_workers
        worker_args = worker_args or {}
        self._workers = [
            worker_class(seed=seed, max_episode_length=max_episode_length,
                         worker_number=i, **worker_args)
            for i in range(n_workers)
        ]
        for w, a, e in zip(self._workers, self._spread(agents),
                           self._spread(envs, copy.deepcopy)):
            w.update_agent(a)
            w.update_env(e)
        self.total_env_steps = 0

    def _spread(self, objs, preprocess=lambda v: v):
        """``worker_factory.py:68-95``."""
        if isinstance(objs, list):
            if len(objs) != self.n_workers:
                raise ValueError(
                    'Length of list doesn\'t match number of workers')
            return [preprocess(o) for o in objs]
        return [preprocess(objs) for _ in range(self.n_workers)]

    def _update_workers(self, agent_update, env_update):
        for w, a, e in zip(self._workers, self._spread(agent_update),
                           self._spread(env_update, copy.deepcopy)):
            w.update_agent(a)
            w.update_env(e)

    def obtain_samples(self, itr, num_samples, agent_update, env_update=None):
        """``local_sampler.py:134-166``."""
        self._update_workers(agent_update, env_update)
        batches, done = [], 0
        while True:
            for w in self._workers:
                b = w.rollout()
                done += len(b.actions)
                batches.append(b)
                if done >= num_samples:
                    out = OracleEpisodeBatch.concatenate(*batches)
                    self.total_env_steps += int(sum(out.lengths))
                    return out

    def obtain_exact_episodes(self, n_eps_per_worker, agent_update,
                              env_update=None):
        """``local_sampler.py:168-200``: worker order, not completion order."""
        self._update_workers(agent_update, env_update)
        batches = [w.rollout() for w in self._workers
                   for _ in range(n_eps_per_worker)]
        out = OracleEpisodeBatch.concatenate(*batches)
        self.total_env_steps += int(sum(out.lengths))
        return out

    def shutdown_worker(self):
        for w in self._workers:
            w.shutdown()


class NormalizedObs:
    """``envs/normalized_env.py:118-124,134-151`` observation EMA (float64).

    The running mean is updated first, the variance uses the *new* mean, and
    the observation is normalised by the updated statistics.
    """

    def __init__(self, dim, alpha=0.001):
        self.alpha = alpha
        self.mean = np.zeros(dim)
        self.var = np.ones(dim)

    def __call__(self, obs):
        obs = np.asarray(obs).reshape(-1)
        a = self.alpha
        self.mean = (1 - a) * self.mean + a * obs
        self.var = (1 - a) * self.var + a * np.square(obs - self.mean)
        return (obs - self.mean) / (np.sqrt(self.var) + 1e-8)
