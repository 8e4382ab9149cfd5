"""Batched environments behind the GPU sampler.

garage steps one ``Environment`` object per env in a Python loop
(``sampler/vec_worker.py:185-197``; contract ``_environment.py:237-276``).  Here
an environment *batch* is one object that advances all ``n_envs`` members with
a single call and keeps its state in HBM:

    reset_all()            -> ``obs`` (n, ldo) holds every first observation
    step_all(actions)      -> ``reward`` (n,), ``step_type`` (n,) uint8 and
                              ``next_obs`` (n, ldo) (the true next observation,
                              terminal ones included)
    reset_where(done)      -> members with ``done != 0`` start a new episode;
                              their first observation overwrites ``next_obs``

``SyntheticVecEnv`` is the benchmark workload of BASELINE.json (HIP kernels,
Philox-keyed).  ``HostVecEnv`` adapts a list of ordinary per-env objects (any
``garage.Environment``-like with ``reset``/``step``) so existing CPU simulators
still feed the device-resident update path.
"""
import ctypes as C

import numpy as np
import torch

from garage_amd import _lib
from garage_amd._dtypes import Box, Discrete, EnvSpec, StepType, is_discrete
from garage_amd._lib import call, dptr, stream_ptr
from garage_amd.engine import require_gpu, round4


class VecEnv:
    """Base class / protocol of a device-batched environment."""

    n_envs = 0
    spec = None

    @property
    def obs_dim(self):
        return self.spec.observation_space.flat_dim

    @property
    def act_width(self):
        """Columns of the action matrix handed to :meth:`step_all`."""
        if is_discrete(self.spec.action_space):
            return 1
        return self.spec.action_space.flat_dim

    def _alloc(self, device):
        n, ldo = self.n_envs, round4(self.obs_dim)
        self.device = device
        self.obs = torch.zeros(n, ldo, dtype=torch.float32, device=device)
        self.next_obs = torch.zeros(n, ldo, dtype=torch.float32, device=device)
        self.reward = torch.zeros(n, dtype=torch.float32, device=device)
        self.step_type = torch.zeros(n, dtype=torch.uint8, device=device)

    def advance(self):
        """Make ``next_obs`` the current observation (buffer swap)."""
        self.obs, self.next_obs = self.next_obs, self.obs

    def hold(self):
        """Carry the current observations over as the next ones (before a
        partial ``reset_where``: rows that are not reset keep their state)."""
        self.next_obs.copy_(self.obs)

    def reset_all(self):
        raise NotImplementedError

    def step_all(self, actions):
        raise NotImplementedError

    def reset_where(self, done):
        raise NotImplementedError

    def close(self):
        pass


class SyntheticVecEnv(VecEnv):
    """``n_envs`` synthetic environments stepped by one HIP kernel.

    Observation ``t`` of episode ``e`` of env ``i``: ``obs_dim`` unit-variance
    uniforms from Philox4x32-10 keyed ``(seed; env_id0 + i, e, t)``; reward: a
    Philox value plus ``0.1 * <clip(a), obs>`` (continuous) or ``0.1 * obs[a]``
    (discrete); episode length ``L ~ U{min_len..max_episode_length}``, ending
    TIMEOUT when ``L == max_episode_length`` else TERMINAL.  ``oracle/envs.py``
    holds the bit-identical per-env CPU twin used by the parity tests.
    """

    def __init__(self, n_envs, obs_dim, act_dim, max_episode_length, *,
                 min_len=None, seed=0, discrete=False, env_id0=0, device=None,
                 action_bounds=None):
        self.n_envs = int(n_envs)
        self._obs_dim = int(obs_dim)
        self._act_dim = int(act_dim)
        self.discrete = bool(discrete)
        self.max_episode_length = int(max_episode_length)
        self.min_len = (self.max_episode_length
                        if min_len is None else int(min_len))
        self.seed = int(seed)
        self.env_id0 = int(env_id0)
        # action_bounds = (low, high): the declared Box of a continuous action
        # space (what ``normalize`` rescales to); the dynamics do not change
        lo, hi = (-np.inf, np.inf) if action_bounds is None else action_bounds
        act_space = (Discrete(act_dim) if discrete else Box(
            lo, hi, (act_dim, )) if np.isscalar(lo) else Box(lo, hi))
        self.spec = EnvSpec(Box(-np.inf, np.inf, (obs_dim, )), act_space,
                            max_episode_length=self.max_episode_length)
        self._init_device(device)

    def _init_device(self, device):
        device = device or require_gpu()
        self._alloc(device)
        n = self.n_envs
        self._episode = torch.full((n, ), -1, dtype=torch.int32, device=device)
        self._t = torch.zeros(n, dtype=torch.int32, device=device)
        self._len = torch.zeros(n, dtype=torch.int32, device=device)
        e = _lib.SynthEnv()
        e.n, e.env_id0 = n, self.env_id0
        e.obs_dim, e.act_dim = self._obs_dim, self._act_dim
        e.discrete = int(self.discrete)
        e.min_len, e.max_len = self.min_len, self.max_episode_length
        e.seed = self.seed
        e.episode = self._episode.data_ptr()
        e.t = self._t.data_ptr()
        e.len = self._len.data_ptr()
        self._c = e

    def reset_all(self):
        call('ga_synth_env_reset', C.byref(self._c), None, dptr(self.obs),
             self.obs.stride(0), stream_ptr())

    def step_all(self, actions):
        call('ga_synth_env_step', C.byref(self._c), dptr(actions),
             actions.stride(0), dptr(self.obs), dptr(self.next_obs),
             self.obs.stride(0), dptr(self.reward), dptr(self.step_type),
             stream_ptr())

    def reset_where(self, done):
        call('ga_synth_env_reset', C.byref(self._c), dptr(done),
             dptr(self.next_obs), self.next_obs.stride(0), stream_ptr())

    # -- snapshot support (trainer.py:263-293 pickles algo + env) -----------
    def __getstate__(self):
        state = {
            k: v for k, v in self.__dict__.items()
            if not torch.is_tensor(v) and k not in ('_c', 'device')
        }
        state['_saved'] = {
            k: getattr(self, k).cpu().numpy()
            for k in ('_episode', '_t', '_len', 'obs')
        }
        return state

    def __setstate__(self, state):
        saved = state.pop('_saved')
        self.__dict__.update(state)
        self._init_device(None)
        for k, v in saved.items():
            getattr(self, k).copy_(torch.from_numpy(v))


class NormalizedVecEnv(VecEnv):
    """``garage.envs.normalize`` for a device batch (``envs/normalized_env.py``).

    Wraps any :class:`VecEnv`; keeps one float64 moving mean / variance per
    member env (the reference deep-copies a ``NormalizedEnv`` per env, so the
    statistics are per env there too) and normalises every observation the
    policy sees -- first observations, next observations and terminal ones --
    with the statistics updated *by that observation* (``:144-147``).  Rewards
    are scaled by ``scale_reward`` and optionally normalised the same way.
    Actions of a ``Box`` action space whose bounds pass the reference's test
    (``:92-94``: no ``-inf`` in ``low`` **or in ``high``** -- its check of the
    upper bound compares with ``-inf`` too, which is kept) are rescaled from
    ``[-expected_action_scale, expected_action_scale]`` to ``[low, high]`` and
    clipped before the wrapped env steps on them (``:90-100``); the batch keeps
    the policy's own actions, as ``EnvStep.action`` does (``:109``).  Same
    keywords, order and defaults as ``garage.envs.normalize``.
    """

    def __init__(self, env, scale_reward=1., normalize_obs=False,
                 normalize_reward=False, expected_action_scale=1.,
                 flatten_obs=True, obs_alpha=0.001, reward_alpha=0.001):
        self._env = env
        self.n_envs = env.n_envs
        self.spec = env.spec
        self._scale_reward = float(scale_reward)
        self._normalize_reward = bool(normalize_reward)
        self._normalize_obs = bool(normalize_obs)
        self._expected_action_scale = float(expected_action_scale)
        # observations of a device batch are flat rows either way
        self._flatten_obs = bool(flatten_obs)
        self._obs_alpha = float(obs_alpha)
        self._reward_alpha = float(reward_alpha)
        dev = env.device
        self.device = dev
        self._act_low = self._act_high = self._scaled = None
        space = env.spec.action_space
        if isinstance(space, Box):
            lb = np.asarray(space.low, dtype=np.float32).reshape(-1)
            ub = np.asarray(space.high, dtype=np.float32).reshape(-1)
            if np.all(lb != -np.inf) and np.all(ub != -np.inf):
                self._act_low = torch.from_numpy(lb.copy()).to(dev)
                self._act_high = torch.from_numpy(ub.copy()).to(dev)
                self._scaled = torch.zeros(self.n_envs, round4(lb.size),
                                           dtype=torch.float32, device=dev)
        n, O = self.n_envs, env.obs_dim
        self._obs_mean = torch.zeros(n, O, dtype=torch.float64, device=dev)
        self._obs_var = torch.ones(n, O, dtype=torch.float64, device=dev)
        self._reward_mean = torch.zeros(n, dtype=torch.float64, device=dev)
        self._reward_var = torch.ones(n, dtype=torch.float64, device=dev)
        # the wrapped env keeps its own (raw) observations -- its dynamics must
        # not see the normalised ones, exactly as the inner env of the
        # reference's wrapper does not; the policy reads these
        if self._normalize_obs:
            self._obs = torch.zeros_like(env.obs)
            self._next_obs = torch.zeros_like(env.next_obs)

    # buffers under the protocol names: normalised copies when observations are
    # normalised, else the wrapped env's own
    obs = property(lambda self: self._obs if self._normalize_obs
                   else self._env.obs)
    next_obs = property(lambda self: self._next_obs if self._normalize_obs
                        else self._env.next_obs)
    reward = property(lambda self: self._env.reward)
    step_type = property(lambda self: self._env.step_type)
    env_id0 = property(lambda self: getattr(self._env, 'env_id0', 0))

    @property
    def last_env_infos(self):
        """Pass-through of a wrapped CPU env batch's per-step ``env_info``."""
        return getattr(self._env, 'last_env_infos', None)

    def pop_finished_episode_infos(self):
        pop = getattr(self._env, 'pop_finished_episode_infos', None)
        return pop() if pop is not None else []

    def advance(self):
        self._env.advance()
        if self._normalize_obs:
            self._obs, self._next_obs = self._next_obs, self._obs

    def hold(self):
        self._env.hold()
        if self._normalize_obs:
            self._next_obs.copy_(self._obs)

    def _norm(self, src, dst, mask=None):
        if self._normalize_obs:
            call('ga_obs_normalize_from_f64', self.n_envs, self.obs_dim,
                 dptr(src), dptr(dst), src.stride(0), dptr(self._obs_mean),
                 dptr(self._obs_var), self._obs_alpha, dptr(mask),
                 stream_ptr())

    def reset_all(self):
        self._env.reset_all()
        self._norm(self._env.obs, self.obs)

    def step_all(self, actions):
        if self._act_low is not None:
            A = self._act_low.numel()
            call('ga_action_rescale_f32', self.n_envs, A, dptr(actions),
                 actions.stride(0), dptr(self._act_low), dptr(self._act_high),
                 self._expected_action_scale, dptr(self._scaled),
                 self._scaled.stride(0), stream_ptr())
            actions = self._scaled
        self._env.step_all(actions)
        self._norm(self._env.next_obs, self.next_obs)
        if self._normalize_reward or self._scale_reward != 1.0:
            call('ga_reward_normalize_f64', self.n_envs,
                 dptr(self._env.reward), dptr(self._reward_mean),
                 dptr(self._reward_var), self._reward_alpha,
                 self._scale_reward, int(self._normalize_reward),
                 stream_ptr())

    def reset_where(self, done):
        self._env.reset_where(done)
        self._norm(self._env.next_obs, self.next_obs, done)

    def norm_args(self):
        """``ga_norm_args`` for the fused env-step kernel (raw buffers are
        filled in by the caller)."""
        from garage_amd import _lib
        a = _lib.NormArgs()
        a.normalize_obs = int(self._normalize_obs)
        a.normalize_reward = int(self._normalize_reward)
        a.obs_mean, a.obs_var = (self._obs_mean.data_ptr(),
                                 self._obs_var.data_ptr())
        a.obs_alpha = self._obs_alpha
        a.reward_mean, a.reward_var = (self._reward_mean.data_ptr(),
                                       self._reward_var.data_ptr())
        a.reward_alpha, a.reward_scale = (self._reward_alpha,
                                          self._scale_reward)
        if self._act_low is not None:
            a.act_low, a.act_high = (self._act_low.data_ptr(),
                                     self._act_high.data_ptr())
            a.expected_action_scale = self._expected_action_scale
            a.scaled_action = self._scaled.data_ptr()
        return a

    def close(self):
        self._env.close()


class HostVecEnv(VecEnv):
    """Adapter: a list of per-env CPU objects behind the batched protocol.

    Each member follows garage's ``Environment`` contract
    (``_environment.py:237-276``): ``reset() -> (obs, episode_info)`` and
    ``step(action) -> EnvStep`` with ``.reward``, ``.observation``,
    ``.step_type``.  Stepping stays a host loop (that is what those simulators
    are).  Per step the actions come down through a pinned buffer (the one
    stream synchronisation of the step: the simulators need them), the
    simulators write observations / rewards / step types straight into numpy
    views of pinned staging buffers, and those go up with asynchronous copies
    on the worker's stream -- the next step's synchronisation is what makes the
    staging buffers reusable.  Everything downstream stays on the device.
    """

    def __init__(self, envs, spec=None, device=None):
        self.envs = list(envs)
        self.n_envs = len(self.envs)
        self.spec = spec if spec is not None else self.envs[0].spec
        self._alloc(device or require_gpu())
        n, ldo = self.n_envs, self.obs.shape[1]
        self._h_obs = torch.zeros(n, ldo, dtype=torch.float32).pin_memory()
        self._h_rew = torch.zeros(n, dtype=torch.float32).pin_memory()
        self._h_st = torch.zeros(n, dtype=torch.uint8).pin_memory()
        self._h_done = torch.zeros(n, dtype=torch.uint8).pin_memory()
        self._h_act = None  # pinned, sized at the first step
        # numpy views of the pinned buffers: the simulators' outputs land in
        # page-locked memory without a per-element tensor operation
        self._np_obs = self._h_obs.numpy()
        self._np_rew = self._h_rew.numpy()
        self._np_st = self._h_st.numpy()
        self.discrete = is_discrete(self.spec.action_space)
        self._obs_discrete = is_discrete(self.spec.observation_space)
        # the last step's ``EnvStep.env_info`` per env (None when every env
        # reported an empty dict); the worker files them per rollout column and
        # packs them into ``EpisodeBatch.env_infos`` (vec_worker.py:192-193)
        self.last_env_infos = None
        # ``reset()[1]`` of the episode each env is in (default_worker.py:94-96),
        # and -- since the last call of pop_finished_episode_infos -- those of the
        # episodes that ended, as (env index, episode_info)
        self._episode_info = [{} for _ in range(n)]
        self._finished = []

    def _put_obs(self, i, obs):
        # discrete observation spaces are seen one-hot by the networks, as
        # ``observation_space.flatten`` makes them for garage's policies
        # (torch/policies/stochastic_policy.py:70-74)
        row = self._np_obs[i]
        if self._obs_discrete:
            row[:] = 0.0
            row[int(obs)] = 1.0
        else:
            flat = np.asarray(obs, dtype=np.float32).reshape(-1)
            row[:flat.shape[0]] = flat

    def _reset_member(self, i):
        obs, info = self.envs[i].reset()
        self._put_obs(i, obs)
        self._episode_info[i] = dict(info or {})

    def reset_all(self):
        torch.cuda.current_stream().synchronize()  # staging buffers are free
        for i in range(self.n_envs):
            self._reset_member(i)
        self._finished = []
        self.obs.copy_(self._h_obs, non_blocking=True)

    def step_all(self, actions):
        if self._h_act is None or self._h_act.shape != actions.shape:
            self._h_act = torch.zeros(actions.shape,
                                      dtype=torch.float32).pin_memory()
        self._h_act.copy_(actions, non_blocking=True)
        # the one synchronisation of the step (it also retires the previous
        # step's uploads from the staging buffers written below)
        torch.cuda.current_stream().synchronize()
        acts = self._h_act.numpy()
        infos, any_info = [], False
        width = self.act_width
        for i, env in enumerate(self.envs):
            a = int(acts[i, 0]) if self.discrete else acts[i, :width]
            es = env.step(a)
            self._put_obs(i, es.observation)
            self._np_rew[i] = es.reward
            self._np_st[i] = int(es.step_type)
            info = getattr(es, 'env_info', None) or {}
            any_info = any_info or bool(info)
            infos.append(info)
        self.last_env_infos = infos if any_info else None
        self.next_obs.copy_(self._h_obs, non_blocking=True)
        self.reward.copy_(self._h_rew, non_blocking=True)
        self.step_type.copy_(self._h_st, non_blocking=True)

    def reset_where(self, done):
        self._h_done.copy_(done, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        idx = np.nonzero(self._h_done.numpy())[0]
        if idx.size == 0:
            return
        # the staging buffer still holds the rows just uploaded to next_obs
        # unless somebody else wrote that tensor (hold() after advance()):
        # fetch them so that rows which are not reset keep their state
        self._h_obs.copy_(self.next_obs)
        for i in idx:
            self._finished.append((int(i), self._episode_info[int(i)]))
            self._reset_member(int(i))
        self.next_obs.copy_(self._h_obs, non_blocking=True)

    def pop_finished_episode_infos(self):
        """``(env index, episode_info)`` of the episodes that ended since the
        last call, in the order they were reset."""
        out, self._finished = self._finished, []
        return out

    def close(self):
        for env in self.envs:
            env.close()


__all__ = ['VecEnv', 'SyntheticVecEnv', 'HostVecEnv', 'StepType']
