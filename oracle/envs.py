"""Oracle: per-environment CPU objects that follow garage's Environment API.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  The interface restated is
``_environment.py:85-143,237-276`` (``reset() -> (obs, episode_info)``,
``step(a) -> EnvStep``; ``EnvStep.last`` / ``.terminal``) with step types from
``_dtypes.py:42-68``.

``SyntheticEnv`` is the CPU twin of the batched device environment
``garage_amd.envs.SyntheticVecEnv``: both derive every observation, reward and
episode length from Philox4x32-10 keyed by ``(seed; env_id, episode, t,
stream)``, with integer -> fp32 conversions that are exact, so the two produce
bit-identical streams (that is what makes rollout parity testable without a
shared stateful RNG).  Philox is the published Random123 algorithm
(Salmon et al., SC'11); this is an independent numpy statement of it.
"""
import numpy as np

from oracle.batch import StepType

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)
_SQRT3 = np.float32(1.7320508)

STREAM_OBS = 0
STREAM_REWARD = 1
STREAM_LENGTH = 2
STREAM_ACTION = 3


def philox4x32(c0, c1, c2, c3, seed):
    """Vectorised Philox4x32-10.  Counters broadcast; returns 4 uint32 arrays."""
    c0, c1, c2, c3 = np.broadcast_arrays(
        *[np.asarray(c, dtype=np.uint64) for c in (c0, c1, c2, c3)])
    c0, c1, c2, c3 = c0.copy(), c1.copy(), c2.copy(), c3.copy()
    k0 = int(seed) & 0xFFFFFFFF
    k1 = (int(seed) >> 32) & 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0), lo1,
                          hi0 ^ c3 ^ np.uint64(k1), lo0)
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def u32_to_unit_variance(u):
    """uint32 -> fp32 uniform on [-sqrt3, sqrt3): exact in fp32 on any device."""
    f = (u >> np.uint32(8)).astype(np.float32) * np.float32(2.0**-23)
    return (f - np.float32(1.0)) * _SQRT3


def synthetic_values(seed, env_id, episode, t, stream, count):
    """``count`` fp32 values for one (env, episode, t, stream) cell."""
    blocks = (count + 3) // 4
    blk = np.arange(blocks, dtype=np.uint64)
    c3 = (np.uint64(stream) << np.uint64(16)) | blk
    r = philox4x32(env_id, episode, t, c3, seed)
    vals = np.stack(r, axis=-1).reshape(-1)[:count]
    return u32_to_unit_variance(vals)


def action_noise(seed, env_ids, step, act_dim):
    """The standard-normal draws the device policy step adds to the means of
    step ``step`` (global step counter of the worker) when it is not teacher
    forced: per env, Philox blocks ``(env, step, b, STREAM_ACTION << 16)`` turned
    into pairs by Box-Muller, ``u = ((x >> 8) + 0.5) 2^-24``.  This is the
    build's own stream (the reference draws from torch's global RNG,
    ``stochastic_policy.py:85``); logs and sincos are the host's, so the device
    agrees to rounding (1e-6), not bit for bit.  Returns ``(len(env_ids),
    act_dim)`` float32."""
    env_ids = np.asarray(env_ids, dtype=np.uint64)
    blocks = (act_dim + 3) // 4
    out = np.zeros((env_ids.size, 4 * blocks), np.float32)
    for b in range(blocks):
        r = philox4x32(env_ids, int(step) & 0xFFFFFFFF, b,
                       np.uint64(STREAM_ACTION) << np.uint64(16), seed)
        u = [((x >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) *
             np.float32(2.0**-24) for x in r]
        for pair in range(2):
            a, c = u[2 * pair], u[2 * pair + 1]
            rad = np.sqrt(np.float32(-2.0) * np.log(a))
            ang = np.float32(6.28318530717958647692) * c
            out[:, 4 * b + 2 * pair] = rad * np.cos(ang)
            out[:, 4 * b + 2 * pair + 1] = rad * np.sin(ang)
    return out[:, :act_dim]


def synthetic_length(seed, env_id, episode, min_len, max_len):
    """Episode length in ``[min_len, max_len]`` (== max_len when equal)."""
    if min_len >= max_len:
        return max_len
    r = philox4x32(env_id, episode, 0, np.uint64(STREAM_LENGTH) << np.uint64(16),
                   seed)[0]
    span = max_len - min_len + 1
    return int(min_len + int(r) % span)


class EnvStepLite:
    """The fields of ``garage.EnvStep`` the workers read."""

    def __init__(self, action, reward, observation, env_info, step_type):
        self.action = action
        self.reward = reward
        self.observation = observation
        self.env_info = env_info
        self.step_type = step_type

    @property
    def terminal(self):
        return self.step_type == StepType.TERMINAL

    @property
    def timeout(self):
        return self.step_type == StepType.TIMEOUT

    @property
    def last(self):
        return self.step_type in (StepType.TERMINAL, StepType.TIMEOUT)


class SyntheticEnv:
    """One environment of the synthetic benchmark family.

    * observation at step ``t`` of episode ``e``: ``obs_dim`` Philox values,
      unit-variance uniform;
    * reward: one Philox value ``+ 0.1 * sum_j clip(a_j,-1,1) * obs_j`` over
      ``j < min(A, O)`` (continuous; left-to-right fp32, separate multiply and
      add) or ``+ 0.1 * obs[a]`` (discrete);
    * an episode ends after ``L ~ U{min_len..max_len}`` steps, TIMEOUT when
      ``L == max_episode_length`` else TERMINAL.
    """

    def __init__(self, env_id, obs_dim, act_dim, max_episode_length, *,
                 min_len=None, seed=0, discrete=False):
        self.env_id = int(env_id)
        self.obs_dim = obs_dim
        self.act_dim = act_dim
        self.max_episode_length = max_episode_length
        self.min_len = max_episode_length if min_len is None else min_len
        self.seed = seed
        self.discrete = discrete
        self._episode = -1
        self._t = None
        self._obs = None

    def _table(self, steps, stream, count):
        """Values for t = 0..steps-1 in one vectorised Philox call."""
        blocks = (count + 3) // 4
        t = np.arange(steps, dtype=np.uint64)[:, None]
        blk = np.arange(blocks, dtype=np.uint64)[None, :]
        c3 = (np.uint64(stream) << np.uint64(16)) | blk
        r = philox4x32(self.env_id, self._episode, t, c3, self.seed)
        vals = np.stack(r, axis=-1).reshape(steps, -1)[:, :count]
        return u32_to_unit_variance(vals)

    def reset(self):
        self._episode += 1
        self._t = 0
        self._len = synthetic_length(self.seed, self.env_id, self._episode,
                                     self.min_len, self.max_episode_length)
        # The whole episode's cells in two calls (same values as cell-by-cell).
        self._obs_tab = self._table(self._len + 1, STREAM_OBS, self.obs_dim)
        self._rew_tab = self._table(self._len, STREAM_REWARD, 1)[:, 0]
        self._obs = self._obs_tab[0]
        return self._obs.copy(), {}

    def step(self, action):
        if self._t is None:
            raise RuntimeError('reset() must be called before step()!')
        noise = self._rew_tab[self._t]
        if self.discrete:
            shaped = self._obs[int(action) % self.obs_dim]
        else:
            a = np.clip(np.asarray(action, dtype=np.float32), -1.0, 1.0)
            shaped = np.float32(0.0)
            for j in range(min(self.act_dim, self.obs_dim)):
                shaped = np.float32(shaped + np.float32(a[j] * self._obs[j]))
        reward = np.float32(noise + np.float32(np.float32(0.1) * shaped))
        self._t += 1
        self._obs = self._obs_tab[self._t]
        done = self._t >= self._len
        step_type = StepType.get_step_type(self._t, self.max_episode_length,
                                           done)
        if step_type in (StepType.TERMINAL, StepType.TIMEOUT):
            self._t = None
        return EnvStepLite(action, float(reward), self._obs.copy(), {},
                           step_type)

    def close(self):
        pass


class CountingEnv:
    """Deterministic bookkeeping fixture.

    ``obs = [env_id, episode, t]``, ``reward = t`` and episode ``e`` of env
    ``i`` lasts ``lengths[e % len(lengths)]`` steps.
    """

    def __init__(self, env_id, lengths, max_episode_length):
        self.env_id = env_id
        self.lengths = list(lengths)
        self.max_episode_length = max_episode_length
        self._episode = -1
        self._t = None

    def _obs(self):
        return np.array([self.env_id, self._episode, self._t],
                        dtype=np.float32)

    def reset(self):
        self._episode += 1
        self._t = 0
        return self._obs(), {}

    def step(self, action):
        self._t += 1
        L = self.lengths[self._episode % len(self.lengths)]
        step_type = StepType.get_step_type(self._t, self.max_episode_length,
                                           self._t >= L)
        return EnvStepLite(action, float(self._t - 1), self._obs(), {},
                           step_type)

    def close(self):
        pass


class TaskEnv(CountingEnv):
    """``CountingEnv`` that reports multi-task ``env_info`` entries the way the
    reference's multi-task wrappers do (``envs/multi_env_wrapper.py:216-219``
    adds ``task_id`` / ``task_name``; MetaWorld-style envs add ``success``):
    ``success`` turns on at step ``success_at`` of an episode (never if None)."""

    def __init__(self, env_id, lengths, max_episode_length, task_id,
                 task_name=None, success_at=None):
        super().__init__(env_id, lengths, max_episode_length)
        self.task_id = task_id
        self.task_name = task_name
        self.success_at = success_at

    def step(self, action):
        es = super().step(action)
        info = {'task_id': self.task_id,
                'success': bool(self.success_at is not None
                                and self._t >= self.success_at)}
        if self.task_name is not None:
            info['task_name'] = self.task_name
        return EnvStepLite(es.action, es.reward, es.observation, info,
                           es.step_type)


class ActionEchoEnv:
    """Fixture for ``NormalizedEnv``'s action and reward paths
    (``envs/normalized_env.py:90-132,153-164``): remembers every action it is
    stepped with (what the wrapper hands the wrapped env after rescaling and
    clipping), observation ``[t, 2 t, -t] + env_id``, reward
    ``0.25 t + sum(action)`` in float64, fixed-length episodes."""

    def __init__(self, env_id, act_dim, max_episode_length):
        self.env_id = env_id
        self.act_dim = act_dim
        self.max_episode_length = max_episode_length
        self.received = []
        self._t = None

    def _obs(self):
        t = float(self._t)
        return np.array([t, 2 * t, -t], dtype=np.float32) + np.float32(
            self.env_id)

    def reset(self):
        self._t = 0
        return self._obs(), {}

    def step(self, action):
        a = np.array(action, dtype=np.float32).reshape(-1)[:self.act_dim]
        self.received.append(a.copy())
        self._t += 1
        reward = 0.25 * (self._t - 1) + float(np.sum(a.astype(np.float64)))
        step_type = StepType.get_step_type(self._t, self.max_episode_length,
                                           False)
        return EnvStepLite(action, reward, self._obs(), {}, step_type)

    def close(self):
        pass
