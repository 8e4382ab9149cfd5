"""Oracle: step types, the packed episode container, minibatch iterator.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Restates
  * ``_dtypes.py:15-68``    StepType and StepType.get_step_type
  * ``_dtypes.py:455-977``  the parts of EpisodeBatch the hot path reads
                            (packed fields, concatenate, padded views, valids)
  * ``np/optimizers/minibatch_dataset.py:4-35``  BatchDataset
  * ``_functions.py:233-275``  log_performance statistics
"""
import enum

import numpy as np

from oracle.returns import discount_cumsum, pad_batch_array


class StepType(enum.IntEnum):
    """``_dtypes.py:30-39``."""
    FIRST = 0
    MID = 1
    TERMINAL = 2
    TIMEOUT = 3

    @classmethod
    def get_step_type(cls, step_cnt, max_episode_length, done):
        """``_dtypes.py:42-68``: TIMEOUT wins over done; FIRST only at 1."""
        if max_episode_length is not None and step_cnt >= max_episode_length:
            return cls.TIMEOUT
        if done:
            return cls.TERMINAL
        if step_cnt == 1:
            return cls.FIRST
        if step_cnt < 1:
            raise ValueError('Expect step_cnt to be >= 1, but got {} '
                             'instead. Did you forget to call `reset('
                             ')`?'.format(step_cnt))
        return cls.MID


class OracleEpisodeBatch:
    """Packed ``N . [T]`` episode batch (``_dtypes.py:455-527`` layout)."""

    FIELDS = ('observations', 'last_observations', 'actions', 'rewards',
              'step_types', 'lengths')

    def __init__(self, *, observations, last_observations, actions, rewards,
                 step_types, lengths, agent_infos=None, env_infos=None,
                 max_episode_length=None):
        self.observations = np.asarray(observations)
        self.last_observations = np.asarray(last_observations)
        self.actions = np.asarray(actions)
        self.rewards = np.asarray(rewards)
        self.step_types = np.asarray(step_types)
        self.lengths = np.asarray(lengths)
        self.agent_infos = dict(agent_infos or {})
        self.env_infos = dict(env_infos or {})
        self.max_episode_length = max_episode_length
        assert self.observations.shape[0] == int(self.lengths.sum())

    @classmethod
    def concatenate(cls, *batches):
        """``_dtypes.py:592-632``: field-wise ``np.concatenate``."""
        first = batches[0]
        cat = {
            f: np.concatenate([getattr(b, f) for b in batches])
            for f in cls.FIELDS
        }
        infos = {
            k: np.concatenate([b.agent_infos[k] for b in batches])
            for k in first.agent_infos
        }
        einfos = {
            k: np.concatenate([b.env_infos[k] for b in batches])
            for k in first.env_infos
        }
        return cls(agent_infos=infos, env_infos=einfos,
                   max_episode_length=first.max_episode_length, **cat)

    @property
    def padded_observations(self):
        """``_dtypes.py:853-862``."""
        return pad_batch_array(self.observations, self.lengths,
                               self.max_episode_length)

    @property
    def padded_rewards(self):
        """``_dtypes.py:903-912``."""
        return pad_batch_array(self.rewards, self.lengths,
                               self.max_episode_length)

    @property
    def padded_actions(self):
        return pad_batch_array(self.actions, self.lengths,
                               self.max_episode_length)

    @property
    def valids(self):
        """``_dtypes.py:915-923``: 1 where a padded cell holds a real step."""
        return pad_batch_array(np.ones_like(self.rewards), self.lengths,
                               self.max_episode_length)

    def episode_ranges(self):
        start = 0
        for n in self.lengths:
            yield start, start + int(n)
            start += int(n)


class BatchDataset:
    """``np/optimizers/minibatch_dataset.py:4-35``.

    One ``np.random.shuffle`` of the id array at construction and one more
    after every full pass, applied cumulatively to the same array, drawn from
    the *global* numpy RNG; the last minibatch of a pass may be short.
    """

    def __init__(self, inputs, batch_size):
        self._inputs = list(inputs)
        self._batch_size = batch_size
        if batch_size is not None:
            self._ids = np.arange(self._inputs[0].shape[0])
            np.random.shuffle(self._ids)

    @property
    def number_batches(self):
        if self._batch_size is None:
            return 1
        return int(np.ceil(self._inputs[0].shape[0] / self._batch_size))

    def iterate(self):
        if self._batch_size is None:
            yield list(self._inputs)
            return
        for k in range(self.number_batches):
            ids = self._ids[k * self._batch_size:(k + 1) * self._batch_size]
            yield [d[ids] for d in self._inputs]
        np.random.shuffle(self._ids)


def minibatch_index_stream(n_samples, batch_size, epochs):
    """All minibatch id arrays one ``OptimizerWrapper.get_minibatch`` yields.

    ``torch/optimizers/optimizer_wrapper.py:31-49``: a fresh ``BatchDataset``
    (one shuffle) and ``epochs`` passes (one shuffle after each).
    """
    if batch_size is None:
        return [None] * epochs
    ids = np.arange(n_samples)
    np.random.shuffle(ids)
    out = []
    nb = int(np.ceil(n_samples / batch_size))
    for _ in range(epochs):
        for k in range(nb):
            out.append(ids[k * batch_size:(k + 1) * batch_size].copy())
        np.random.shuffle(ids)
    return out


def performance_stats(batch, discount):
    """The numbers ``log_performance`` records (``_functions.py:233-275``)."""
    first_returns, undiscounted, termination, success = [], [], [], []
    for start, stop in batch.episode_ranges():
        rew = batch.rewards[start:stop]
        first_returns.append(discount_cumsum(rew, discount)[0])
        undiscounted.append(sum(rew))
        st = batch.step_types[start:stop]
        termination.append(float(any(int(s) == StepType.TERMINAL for s in st)))
        if 'success' in batch.env_infos:
            success.append(float(batch.env_infos['success'][start:stop].any()))
    stats = {
        'NumEpisodes': len(first_returns),
        'AverageDiscountedReturn': np.mean(first_returns),
        'AverageReturn': np.mean(undiscounted),
        'StdReturn': np.std(undiscounted),
        'MaxReturn': np.max(undiscounted),
        'MinReturn': np.min(undiscounted),
        'TerminationRate': np.mean(termination),
    }
    if success:
        stats['SuccessRate'] = np.mean(success)
    return stats, undiscounted


def select_episodes(batch, which):
    """The episodes ``which`` (indices) of ``batch`` as a new batch, in that order
    (``EpisodeBatch.split`` + ``concatenate``, ``_dtypes.py:592-674``)."""
    ranges = list(batch.episode_ranges())
    rows = np.concatenate([np.arange(*ranges[e]) for e in which])
    which = np.asarray(which)
    return OracleEpisodeBatch(
        observations=batch.observations[rows],
        last_observations=batch.last_observations[which],
        actions=batch.actions[rows], rewards=batch.rewards[rows],
        step_types=batch.step_types[rows], lengths=batch.lengths[which],
        agent_infos={k: v[rows] for k, v in batch.agent_infos.items()},
        env_infos={k: v[rows] for k, v in batch.env_infos.items()},
        max_episode_length=batch.max_episode_length)


def multitask_performance_stats(itr, batch, discount, name_map=None):
    """What ``log_multitask_performance`` records (``_functions.py:177-230``), as
    ``{prefixed key: value}`` plus the undiscounted returns it returns.

    Episodes are grouped by the FIRST step's ``task_name`` env-info, else by
    ``name_map[task_id]`` (default ``'Task #<id>'``), else ``'__unnamed_task__'``;
    with a ``name_map`` every listed task (and only those) is logged, absent
    ones as NaN rows with ``NumEpisodes = 0``; the whole batch is logged under
    ``Average/``."""
    groups = {}
    for e, (start, _) in enumerate(batch.episode_ranges()):
        name = '__unnamed_task__'
        if 'task_name' in batch.env_infos:
            name = batch.env_infos['task_name'][start]
        elif 'task_id' in batch.env_infos:
            # (_functions.py:204: the map is REPLACED by {} when it was None, so
            # with task ids and no map the per-task rows below are skipped --
            # ``task_names = name_map.values()`` is then empty; pinned by
            # tests/golden/multitask.npz ``ids_nomap``)
            name_map = {} if name_map is None else name_map
            task_id = batch.env_infos['task_id'][start]
            name = name_map.get(task_id, 'Task #{}'.format(task_id))
        groups.setdefault(name, []).append(e)
    names = list(groups) if name_map is None else list(name_map.values())
    out = {}
    for name in names:
        out[name + '/Iteration'] = itr
        if name in groups:
            stats, _ = performance_stats(select_episodes(batch, groups[name]),
                                         discount)
            for k, v in stats.items():
                out[name + '/' + k] = v
        else:
            out[name + '/NumEpisodes'] = 0
            for k in ('AverageDiscountedReturn', 'AverageReturn', 'StdReturn',
                      'MaxReturn', 'MinReturn', 'TerminationRate',
                      'SuccessRate'):
                out[name + '/' + k] = np.nan
    stats, undiscounted = performance_stats(batch, discount)
    out['Average/Iteration'] = itr
    for k, v in stats.items():
        out['Average/' + k] = v
    return out, undiscounted
