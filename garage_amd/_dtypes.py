"""Data types of the hot path: spaces, EnvSpec, StepType, EpisodeBatch.

Mirrors ``garage._dtypes`` / ``garage._environment`` for the fields the
on-policy path touches (``_dtypes.py:15-68,455-977``; ``_environment.py:22-79``)
so that code written against garage's ``EpisodeBatch`` reads ours unchanged.
:class:`DeviceEpisodeBatch` keeps the arrays resident in HBM and materialises
the numpy attributes lazily, on first access.
"""
import enum
import warnings

import numpy as np


class StepType(enum.IntEnum):
    """``_dtypes.py:15-68``."""
    FIRST = 0
    MID = 1
    TERMINAL = 2
    TIMEOUT = 3

    @classmethod
    def get_step_type(cls, step_cnt, max_episode_length, done):
        if max_episode_length is not None and step_cnt >= max_episode_length:
            return StepType.TIMEOUT
        if done:
            return StepType.TERMINAL
        if step_cnt == 1:
            return StepType.FIRST
        if step_cnt < 1:
            raise ValueError('Expect step_cnt to be >= 1, but got {} '
                             'instead. Did you forget to call `reset('
                             ')`?'.format(step_cnt))
        return StepType.MID


_STEP_TYPES = np.array([StepType.FIRST, StepType.MID, StepType.TERMINAL,
                        StepType.TIMEOUT], dtype=object)


class Box:
    """Continuous space (the subset of ``akro.Box`` the path uses)."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            low = np.asarray(low, dtype=dtype)
            high = np.asarray(high, dtype=dtype)
            shape = low.shape
        else:
            low = np.full(shape, low, dtype=dtype)
            high = np.full(shape, high, dtype=dtype)
        self.low, self.high = low, high
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)

    @property
    def flat_dim(self):
        return int(np.prod(self.shape))

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(
            np.all(x >= self.low) and np.all(x <= self.high))

    def flatten(self, x):
        return np.asarray(x).flatten()

    def flatten_n(self, xs):
        xs = np.asarray(xs)
        return xs.reshape((xs.shape[0], -1))

    def unflatten(self, x):
        return np.asarray(x).reshape(self.shape)

    def __repr__(self):
        return 'Box{}'.format(self.shape)

    def __eq__(self, other):
        return (isinstance(other, Box) and self.shape == other.shape
                and np.allclose(self.low, other.low)
                and np.allclose(self.high, other.high))

    def __hash__(self):
        return hash(self.shape)


class Discrete:
    """Discrete space with ``n`` choices (``akro.Discrete`` subset)."""

    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype(np.int64)

    @property
    def flat_dim(self):
        return self.n

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == () and 0 <= int(x) < self.n

    def flatten(self, x):
        """One-hot vector of length ``n`` (``akro.Discrete.flatten``)."""
        out = np.zeros(self.n, dtype=np.float32)
        out[int(x)] = 1.0
        return out

    def flatten_n(self, xs):
        xs = np.asarray(xs, dtype=np.int64).reshape(-1)
        out = np.zeros((xs.shape[0], self.n), dtype=np.float32)
        out[np.arange(xs.shape[0]), xs] = 1.0
        return out

    def unflatten(self, x):
        return int(np.nonzero(np.asarray(x))[0][0])

    def __repr__(self):
        return 'Discrete({})'.format(self.n)

    def __eq__(self, other):
        return isinstance(other, Discrete) and self.n == other.n

    def __hash__(self):
        return hash(self.n)


def is_discrete(space):
    return hasattr(space, 'n') and tuple(getattr(space, 'shape', ())) == ()


class EnvSpec:
    """``_environment.py:22-79``."""

    def __init__(self, observation_space, action_space,
                 max_episode_length=None):
        self.observation_space = observation_space
        self.action_space = action_space
        self.max_episode_length = max_episode_length

    def __eq__(self, other):
        return (isinstance(other, EnvSpec)
                and self.observation_space == other.observation_space
                and self.action_space == other.action_space
                and self.max_episode_length == other.max_episode_length)

    def __hash__(self):
        return hash((self.observation_space, self.action_space,
                     self.max_episode_length))

    def __repr__(self):
        return 'EnvSpec({!r}, {!r}, max_episode_length={})'.format(
            self.observation_space, self.action_space, self.max_episode_length)


def _soft_contains(space, element):
    """``_dtypes.py:_space_soft_contains``: exact or flattened membership."""
    if space.contains(element):
        return True
    if hasattr(space, 'flat_dim'):
        return getattr(element, 'shape', None) == (space.flat_dim, ) or (
            getattr(element, 'shape', None) == tuple(space.shape))
    return False


def pad_batch_array(array, lengths, max_length=None):
    """``np/_functions.py:375-406``."""
    lengths = [int(v) for v in lengths]
    assert array.shape[0] == sum(lengths)
    if max_length is None:
        max_length = max(lengths)
    elif max_length < max(lengths):
        warnings.warn('Creating a padded array with longer length than '
                      'requested')
        max_length = max(lengths)
    padded = np.zeros((len(lengths), max_length) + array.shape[1:],
                      dtype=array.dtype)
    start = 0
    for i, n in enumerate(lengths):
        padded[i, :n] = array[start:start + n]
        start += n
    return padded


class EpisodeBatch:
    r"""Packed batch of whole episodes (``_dtypes.py:455-977``).

    Fields and their shapes follow the reference: ``observations``
    :math:`(N \bullet [T], O^*)`, ``last_observations`` :math:`(N, O^*)`,
    ``actions``, ``rewards``, ``step_types`` (object array of
    :class:`StepType`), ``env_infos`` / ``agent_infos`` (dicts of
    :math:`(N \bullet [T], ...)`), ``episode_infos_by_episode`` and the
    integer ``lengths`` :math:`(N,)`.  Construction validates like the
    reference and raises ``ValueError`` with the same messages
    (``_dtypes.py:528-589,1001-1083``).
    """

    def __init__(self, env_spec, episode_infos, observations,
                 last_observations, actions, rewards, env_infos, agent_infos,
                 step_types, lengths):
        if len(lengths.shape) != 1:
            raise ValueError(
                f'lengths has shape {lengths.shape} but must be a ternsor of '
                f'shape (N,)')
        if lengths.dtype.kind not in 'ui':
            raise ValueError(
                f'lengths has dtype {lengths.dtype}, but must have an '
                f'integer dtype')
        n_episodes = len(lengths)
        for key, val in episode_infos.items():
            if not isinstance(val, np.ndarray):
                raise ValueError(
                    f'Entry {key!r} in episode_infos is of type {type(val)!r} '
                    f'but must be of type {np.ndarray!r}')
            if val.shape[0] != n_episodes:
                raise ValueError(
                    f'Entry {key!r} in episode_infos has batch size '
                    f'{val.shape[0]}, but must have batch size '
                    f'{n_episodes} to match the number of episodes')
        if not isinstance(last_observations, np.ndarray):
            raise ValueError(
                f'last_observations is not of type {np.ndarray!r}')
        if last_observations.shape[0] != n_episodes:
            raise ValueError(
                f'last_observations has batch size '
                f'{last_observations.shape[0]} but must have '
                f'batch size {n_episodes} to match the number of episodes')
        if not _soft_contains(env_spec.observation_space,
                              last_observations[0]):
            raise ValueError('last_observations must have the same '
                             'number of entries as there are episodes '
                             f'({n_episodes}) but got data with shape '
                             f'{last_observations[0].shape} entries')
        self.env_spec = env_spec
        self.episode_infos_by_episode = episode_infos
        self.observations = observations
        self.last_observations = last_observations
        self.actions = actions
        self.rewards = rewards
        self.env_infos = env_infos
        self.agent_infos = agent_infos
        self.step_types = step_types
        self.lengths = lengths
        self._validate_steps()

    def _validate_steps(self):
        spec = self.env_spec
        size, size_field = None, None
        for field in ('rewards', 'observations', 'actions', 'step_types'):
            value = getattr(self, field)
            if not isinstance(value, np.ndarray):
                raise ValueError(f'{field} is not of type {np.ndarray!r}')
            if size is None:
                size, size_field = value.shape[0], field
            elif value.shape[0] != size:
                raise ValueError(
                    f'{field} has batch size {value.shape[0]}, but '
                    f'must have batch size {size} '
                    f'to match {size_field}')
            if field == 'observations' and not _soft_contains(
                    spec.observation_space, value[0]):
                raise ValueError(
                    f'Each observation has shape {value[0].shape} '
                    f'but must match the observation_space '
                    f'{spec.observation_space}')
            if field == 'actions' and not _soft_contains(
                    spec.action_space, value[0]):
                raise ValueError(
                    f'Each action has shape {value[0].shape} '
                    f'but must match the action_space '
                    f'{spec.action_space}')
            if field in ('rewards', 'step_types') and value.shape != (size, ):
                raise ValueError(f'{field} has shape {value.shape} '
                                 f'but must have batch size '
                                 f'{size} to match '
                                 f'{size_field}')
        for field in ('agent_infos', 'env_infos'):
            for key, val in getattr(self, field).items():
                if not isinstance(val, (np.ndarray, dict)):
                    raise ValueError(
                        f'Entry {key!r} in {field} is of type {type(val)}'
                        f'but must be {np.ndarray!r} or dict')
                if hasattr(val, 'shape') and val.shape[0] != size:
                    raise ValueError(
                        f'Entry {key!r} in {field} has batch size '
                        f'{val.shape[0]} but must have batch size '
                        f'{size} to match '
                        f'{size_field}')
        if self.step_types.dtype != np.dtype(object):
            raise ValueError(
                f'step_types has dtype {self.step_types.dtype} but must have '
                f'dtype StepType')

    # ``_dtypes.py:592-632``
    @classmethod
    def concatenate(cls, *batches):
        first = batches[0]

        def cat(get):
            return np.concatenate([get(b) for b in batches])

        return EpisodeBatch(
            env_spec=first.env_spec,
            episode_infos={
                k: cat(lambda b, k=k: b.episode_infos_by_episode[k])
                for k in first.episode_infos_by_episode
            },
            observations=cat(lambda b: b.observations),
            last_observations=cat(lambda b: b.last_observations),
            actions=cat(lambda b: b.actions),
            rewards=cat(lambda b: b.rewards),
            env_infos={
                k: cat(lambda b, k=k: b.env_infos[k])
                for k in first.env_infos
            },
            agent_infos={
                k: cat(lambda b, k=k: b.agent_infos[k])
                for k in first.agent_infos
            },
            step_types=cat(lambda b: b.step_types),
            lengths=cat(lambda b: b.lengths))

    def _episode_ranges(self):
        start = 0
        for n in self.lengths:
            yield start, start + int(n)
            start += int(n)

    def split(self):
        """``_dtypes.py:648-674``: one single-episode batch per episode."""
        out = []
        for i, (a, b) in enumerate(self._episode_ranges()):
            out.append(
                EpisodeBatch(
                    env_spec=self.env_spec,
                    episode_infos={
                        k: v[i:i + 1]
                        for k, v in self.episode_infos_by_episode.items()
                    },
                    observations=self.observations[a:b],
                    last_observations=np.asarray([self.last_observations[i]]),
                    actions=self.actions[a:b],
                    rewards=self.rewards[a:b],
                    env_infos={k: v[a:b] for k, v in self.env_infos.items()},
                    agent_infos={
                        k: v[a:b] for k, v in self.agent_infos.items()
                    },
                    step_types=self.step_types[a:b],
                    lengths=np.asarray([self.lengths[i]])))
        return out

    @property
    def next_observations(self):
        """``_dtypes.py:803-816``: observations shifted by one per episode."""
        rows = []
        for i, (a, b) in enumerate(self._episode_ranges()):
            rows.append(
                np.concatenate([self.observations[a + 1:b],
                                self.last_observations[i:i + 1]]))
        return np.concatenate(rows)

    @property
    def episode_infos(self):
        return {
            k: np.repeat(v, [int(n) for n in self.lengths], axis=0)
            for k, v in self.episode_infos_by_episode.items()
        }

    @property
    def padded_observations(self):
        return pad_batch_array(self.observations, self.lengths,
                               self.env_spec.max_episode_length)

    @property
    def padded_actions(self):
        return pad_batch_array(self.actions, self.lengths,
                               self.env_spec.max_episode_length)

    @property
    def padded_rewards(self):
        return pad_batch_array(self.rewards, self.lengths,
                               self.env_spec.max_episode_length)

    @property
    def valids(self):
        return pad_batch_array(np.ones_like(self.rewards), self.lengths,
                               self.env_spec.max_episode_length)

    @property
    def padded_step_types(self):
        return pad_batch_array(self.step_types, self.lengths,
                               self.env_spec.max_episode_length)

    @property
    def padded_next_observations(self):
        """``_dtypes.py:926-934``."""
        return pad_batch_array(self.next_observations, self.lengths,
                               self.env_spec.max_episode_length)

    @property
    def padded_agent_infos(self):
        """``_dtypes.py:948-961``."""
        return {
            k: pad_batch_array(arr, self.lengths,
                               self.env_spec.max_episode_length)
            for k, arr in self.agent_infos.items()
        }

    @property
    def padded_env_infos(self):
        """``_dtypes.py:964-977``."""
        return {
            k: pad_batch_array(arr, self.lengths,
                               self.env_spec.max_episode_length)
            for k, arr in self.env_infos.items()
        }

    @property
    def observations_list(self):
        """``_dtypes.py:877-887``: one ``(T_i, O^*)`` array per episode."""
        return [self.observations[a:b] for a, b in self._episode_ranges()]

    @property
    def actions_list(self):
        """``_dtypes.py:890-900``."""
        return [self.actions[a:b] for a, b in self._episode_ranges()]

    @property
    def terminals(self):
        """``_dtypes.py:381-390`` (inherited from ``TimeStepBatch``): which steps
        are ``StepType.TERMINAL``."""
        return np.array([s == StepType.TERMINAL for s in self.step_types])

    def to_list(self):
        """``_dtypes.py:676-728``: one dictionary per episode."""
        episodes = []
        infos = self.episode_infos
        for i, (a, b) in enumerate(self._episode_ranges()):
            episodes.append({
                # (the reference slices the per-step expansion by the episode
                # index, ``v[i:i + 1]``: kept)
                'episode_infos': {k: v[i:i + 1] for k, v in infos.items()},
                'observations': self.observations[a:b],
                'next_observations': np.concatenate(
                    (self.observations[1 + a:b], [self.last_observations[i]])),
                'actions': self.actions[a:b],
                'rewards': self.rewards[a:b],
                'env_infos': {k: v[a:b] for k, v in self.env_infos.items()},
                'agent_infos': {k: v[a:b]
                                for k, v in self.agent_infos.items()},
                'step_types': self.step_types[a:b],
            })
        return episodes

    @classmethod
    def from_list(cls, env_spec, paths):
        """``_dtypes.py:731-800``: episodes given as dictionaries (observations
        may hold ``T + 1`` rows, or ``next_observations`` may be given; a list of
        ``dones`` stands in for missing step types)."""
        lengths = np.asarray([len(p['rewards']) for p in paths])
        if all(len(p['observations']) == n + 1
               for p, n in zip(paths, lengths)):
            last_observations = np.asarray(
                [p['observations'][-1] for p in paths])
            observations = np.concatenate(
                [p['observations'][:-1] for p in paths])
        else:
            observations = np.concatenate([p['observations'] for p in paths])
            if paths[0].get('next_observations') is not None:
                last_observations = np.asarray(
                    [p['next_observations'][-1] for p in paths])
            else:
                last_observations = np.asarray(
                    [p['observations'][-1] for p in paths])
        stacked = _concat_tensor_dict_list(paths)
        episode_infos = _stack_tensor_dict_list(
            [p['episode_infos'] for p in paths])
        if 'dones' in stacked and 'step_types' not in stacked:
            stacked['step_types'] = np.array(
                [StepType.TERMINAL if d else StepType.MID
                 for d in stacked['dones']], dtype=StepType)
            del stacked['dones']
        return cls(env_spec=env_spec, episode_infos=episode_infos,
                   observations=observations,
                   last_observations=last_observations,
                   actions=stacked['actions'], rewards=stacked['rewards'],
                   env_infos=stacked['env_infos'],
                   agent_infos=stacked['agent_infos'],
                   step_types=stacked['step_types'], lengths=lengths)


def _stack_tensor_dict_list(dicts):
    """``np/_functions.py:236-259``."""
    out = {}
    for k in list(dicts[0].keys()):
        items = [d[k] if k in d else [] for d in dicts]
        out[k] = (_stack_tensor_dict_list(items)
                  if isinstance(dicts[0][k], dict) else np.array(items))
    return out


def _concat_tensor_dict_list(dicts):
    """``np/_functions.py:296-319``."""
    out = {}
    for k in list(dicts[0].keys()):
        items = [d[k] if k in d else [] for d in dicts]
        out[k] = (_concat_tensor_dict_list(items)
                  if isinstance(dicts[0][k], dict) else
                  np.concatenate(items, axis=0))
    return out


def step_types_as_uint8(step_types):
    """``step_types`` (garage's object array of :class:`StepType`, or integers)
    as a uint8 array.  The four enum members are singletons, so an object array
    of them is an array of four distinct pointers: those are compared in bulk
    (CPython: ``id`` is the address) instead of calling ``int()`` a million
    times; anything else takes numpy's element-wise conversion."""
    arr = np.asarray(step_types)
    if arr.dtype != object:
        return arr.astype(np.uint8)
    flat = np.ascontiguousarray(arr).reshape(-1)
    if flat.size == 0:
        return np.zeros(0, dtype=np.uint8)
    try:
        import ctypes
        ptrs = np.ctypeslib.as_array(
            (ctypes.c_size_t * flat.size).from_address(flat.ctypes.data))
        out = np.full(flat.size, 255, dtype=np.uint8)
        for member in StepType:
            out[ptrs == id(member)] = int(member)
        if not (out == 255).any():
            return out
    except (TypeError, ValueError, AttributeError):  # pragma: no cover
        pass
    return flat.astype(np.uint8)


class DeviceEpisodeBatch(EpisodeBatch):
    """An :class:`EpisodeBatch` whose arrays live in HBM.

    The GPU sampler returns this.  Device tensors are exposed as ``*_dev``
    (padded layouts, see ``DESIGN.md``); every numpy attribute of the base
    class is produced on first access with one D2H copy, so garage code that
    reads ``eps.observations`` keeps working, while ``garage_amd``'s PPO never
    leaves the device.  dtypes follow ``VecWorker``: int64
    lengths, float64 rewards, object step types.
    """

    _LAZY = ('observations', 'last_observations', 'actions', 'rewards',
             'step_types', 'agent_infos')

    def __init__(self, env_spec, *, lengths, obs_dev, last_obs_dev,
                 actions_dev, rewards_dev, step_types_dev, ep_off_dev,
                 head_dev=None, head_name='mean', log_std=None,
                 discrete=False, extras=None, env_infos=None,
                 episode_infos=None):
        # deliberately no base-class __init__: nothing to validate on host
        self.env_spec = env_spec
        self.lengths = np.asarray(lengths, dtype='l')
        # (N, ...) arrays: what each episode's ``reset()`` reported
        self.episode_infos_by_episode = dict(episode_infos or {})
        # host arrays of shape (S, ...): what CPU environments reported per step
        self.env_infos = dict(env_infos or {})
        self.obs_dev = obs_dev
        self.last_obs_dev = last_obs_dev
        self.actions_dev = actions_dev
        self.rewards_dev = rewards_dev
        self.step_types_dev = step_types_dev
        self.ep_off_dev = ep_off_dev
        self.head_dev = head_dev
        self._head_name = head_name
        self._log_std = log_std
        self._discrete = discrete
        self.extras = extras or {}
        self._cache = {}

    @property
    def n_samples(self):
        return int(self.step_types_dev.shape[0])

    def _obs_shape(self):
        return tuple(self.env_spec.observation_space.shape) or (
            self.env_spec.observation_space.flat_dim, )

    def __getattr__(self, name):
        if name in DeviceEpisodeBatch._LAZY:
            cache = self.__dict__.setdefault('_cache', {})
            if name not in cache:
                cache[name] = self._materialise(name)
            return cache[name]
        raise AttributeError(name)

    def _materialise(self, name):
        O = self.env_spec.observation_space.flat_dim
        if name in ('observations', 'last_observations'):
            dev = self.obs_dev if name == 'observations' else self.last_obs_dev
            a = dev[:, :O].cpu().numpy()
            if is_discrete(self.env_spec.observation_space):
                # the networks saw one-hot rows; garage's batch holds the states
                return a.argmax(axis=1).astype(np.int64)
            return a.reshape((a.shape[0], ) + self._obs_shape())
        if name == 'actions':
            if self._discrete:
                return self.actions_dev[:, 0].cpu().numpy().astype(np.int64)
            A = self.env_spec.action_space.flat_dim
            return self.actions_dev[:, :A].cpu().numpy()
        if name == 'rewards':
            return self.rewards_dev.cpu().numpy().astype(np.float64)
        if name == 'step_types':
            return _STEP_TYPES[self.step_types_dev.cpu().numpy()]
        if name == 'agent_infos':
            infos = {}
            if self.head_dev is not None:
                A = self.env_spec.action_space.flat_dim
                infos[self._head_name] = self.head_dev[:, :A].cpu().numpy()
                if self._log_std is not None:
                    infos['log_std'] = np.full_like(infos[self._head_name],
                                                    self._log_std)
            return infos
        raise AttributeError(name)

    def to_host(self):
        """A plain, validated :class:`EpisodeBatch` copy on the host."""
        return EpisodeBatch(env_spec=self.env_spec,
                            episode_infos=dict(self.episode_infos_by_episode),
                            observations=self.observations,
                            last_observations=self.last_observations,
                            actions=self.actions, rewards=self.rewards,
                            env_infos=dict(self.env_infos),
                            agent_infos=self.agent_infos,
                            step_types=self.step_types, lengths=self.lengths)
