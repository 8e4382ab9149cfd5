"""GPU sampler and worker with garage's Sampler / Worker plugin surface.

Replaces ``LocalSampler`` + ``VecWorker`` (``sampler/local_sampler.py:13-232``,
``sampler/vec_worker.py:12-228``, ``sampler/worker_factory.py:24-116``,
``sampler/_functions.py:6-40``).  The per-env Python loop becomes a handful of
kernel launches per step over env-major ``(n_envs, Tcap)`` rollout buffers in
HBM; episode bookkeeping (lengths, completion order, step types, the reset of
finished envs, the discard of in-flight episodes at each ``obtain_samples``)
follows the reference exactly and is verified bit for bit against it.
"""
import copy
import warnings
import ctypes as C

import numpy as np
import torch

from garage_amd import _lib
from garage_amd._dtypes import DeviceEpisodeBatch, EpisodeBatch, is_discrete
from garage_amd._lib import call, dptr, stream_ptr
from garage_amd.engine import require_gpu, round4
from garage_amd.envs import HostVecEnv, VecEnv


def _identity(value):
    return value


class WorkerFactory:
    """``sampler/worker_factory.py:24-116`` (picklable worker constructor)."""

    def __init__(self, *, max_episode_length, is_tf_worker=False, seed=None,
                 n_workers=1, worker_class=None, worker_args=None):
        if is_tf_worker:
            raise NotImplementedError('TensorFlow workers are out of scope')
        self.n_workers = n_workers
        self._seed = seed
        self._max_episode_length = max_episode_length
        self._worker_class = worker_class or GpuVecWorker
        self._worker_args = {} if worker_args is None else worker_args

    def prepare_worker_messages(self, objs, preprocess=_identity):
        if isinstance(objs, list):
            if len(objs) != self.n_workers:
                raise ValueError(
                    'Length of list doesn\'t match number of workers')
            return [preprocess(obj) for obj in objs]
        return [preprocess(objs) for _ in range(self.n_workers)]

    def __call__(self, worker_number):
        if worker_number >= self.n_workers:
            raise ValueError('Worker number is too big')
        return self._worker_class(worker_number=worker_number,
                                  seed=self._seed,
                                  max_episode_length=self._max_episode_length,
                                  **self._worker_args)


class EnvUpdate:
    """``sampler/env_update.py:5-33``: a callable ``old_env -> env``; the caller
    uses what it returns and forgets ``old_env``.  The base class keeps it."""

    def __call__(self, old_env=None):
        return old_env


class NewEnvUpdate(EnvUpdate):
    """``env_update.py:36-63``: closes the old environment and constructs a new
    one with ``env_constructor()`` at every update."""

    def __init__(self, env_constructor):
        self._env_constructor = env_constructor

    def __call__(self, old_env=None):
        if old_env:
            old_env.close()
        return self._env_constructor()


class SetTaskUpdate(EnvUpdate):
    """``env_update.py:66-121``: ``set_task(task)`` on an environment of type
    ``env_type``, constructing (and wrapping) one when the old environment is
    missing or of another type."""

    def __init__(self, env_type, task, wrapper_constructor):
        if not isinstance(env_type, type):
            raise ValueError('env_type should be a type, not '
                             f'{type(env_type)!r}')
        self._env_type = env_type
        self._task = task
        self._wrapper_cons = wrapper_constructor

    def _make_env(self):
        env = self._env_type()
        env.set_task(self._task)
        if self._wrapper_cons is not None:
            env = self._wrapper_cons(env, self._task)
        return env

    def __call__(self, old_env=None):
        if old_env is None:
            return self._make_env()
        if not isinstance(getattr(old_env, 'unwrapped', old_env),
                          self._env_type):
            warnings.warn('SetTaskEnvUpdate is closing an environment. This '
                          'may indicate a very slow TaskSampler setup.')
            old_env.close()
            return self._make_env()
        old_env.set_task(self._task)
        return old_env


class ExistingEnvUpdate(EnvUpdate):
    """``env_update.py:124-159``: hands over an environment that already
    exists; the old one is not closed."""

    def __init__(self, env):
        self._env = env

    def __call__(self, old_env=None):
        return self._env

    def __getstate__(self):
        warnings.warn('ExistingEnvUpdate is generally not the most efficient '
                      'method of transmitting environments to other '
                      'processes.')
        return self.__dict__


def _is_environment(obj):
    return hasattr(obj, 'step') and hasattr(obj, 'reset')


def _apply_env_update(old_env, env_update):
    """``sampler/_functions.py:6-40``: ``(env, updated)``; ``None`` keeps the
    environment, an :class:`EnvUpdate` is called on it, an environment replaces
    it (the old one is closed), anything else is a ``TypeError``."""
    if env_update is None:
        return old_env, False
    if isinstance(env_update, EnvUpdate):
        return env_update(old_env), True
    if _is_environment(env_update):
        if old_env is not None:
            old_env.close()
        return env_update, True
    raise TypeError('Unknown environment update type.')


def _copy_env(env):
    """``copy.deepcopy`` for per-env objects; device batches are shared."""
    return env if isinstance(env, VecEnv) else copy.deepcopy(env)


class GpuVecWorker:
    """``VecWorker`` whose ``n_envs`` environments live on the GPU.

    ``update_env`` accepts a :class:`~garage_amd.envs.VecEnv` (used as is), a
    list of exactly ``n_envs`` per-env objects or a single per-env object that
    is deep-copied ``n_envs`` times (``vec_worker.py:75-105``) -- the latter two
    are wrapped in :class:`~garage_amd.envs.HostVecEnv`.
    """

    DEFAULT_N_ENVS = 8

    def __init__(self, *, seed, max_episode_length, worker_number,
                 n_envs=DEFAULT_N_ENVS, noise_fn=None, store_agent_infos=True,
                 fused_policy_step=True):
        self._use_fused = bool(fused_policy_step)
        self._seed = seed
        self._max_episode_length = max_episode_length
        self._worker_number = worker_number
        self._n_envs = n_envs
        self._noise_fn = noise_fn  # test hook: step -> (n, A) noise tensor
        self._store_infos = store_agent_infos
        self.agent = None
        self.env = None
        self._needs_agent_reset = True
        self._needs_env_reset = True
        self._ep_t = None
        self._global_step = 0
        self._pending = []  # API-compat path: completed episode batches
        self.device = require_gpu()
        if seed is not None:
            # default_worker.py:50-53: seed the *global* RNGs per worker
            import random
            s = seed + worker_number
            random.seed(s)
            np.random.seed(s)
            torch.manual_seed(s)

    # -- updates --------------------------------------------------------------
    def update_agent(self, agent_update):
        """``default_worker.py:55-69`` + ``vec_worker.py:61-73``."""
        if isinstance(agent_update, (dict, tuple, np.ndarray)):
            self.agent.set_param_values(agent_update)
        elif agent_update is not None:
            self.agent = agent_update
        self._needs_agent_reset = True

    def update_env(self, env_update):
        """``vec_worker.py:75-105``.  A :class:`~garage_amd.envs.VecEnv` replaces
        the batch as a whole (so does a plain callable ``VecEnv -> VecEnv``, for
        device-resident batches); a list holds one update per environment --
        ``None``, an environment or an :class:`EnvUpdate` each; a single
        environment or :class:`EnvUpdate` is deep-copied ``n_envs`` times."""
        if env_update is None:
            return
        n = self._n_envs
        if isinstance(env_update, VecEnv) or (
                callable(env_update) and not isinstance(env_update, EnvUpdate)
                and not _is_environment(env_update)):
            new_env = (env_update if isinstance(env_update, VecEnv) else
                       env_update(self.env))
            if new_env.n_envs != n:
                raise ValueError('If separate environments are passed for '
                                 'each worker, there must be exactly n_envs '
                                 '({}) environments, but received {} '
                                 'environments.'.format(n, new_env.n_envs))
            if self.env is not None and self.env is not new_env:
                self.env.close()
            self.env = new_env
            self._needs_env_reset = True
            return
        if isinstance(env_update, list):
            if len(env_update) != n:
                raise ValueError('If separate environments are passed for '
                                 'each worker, there must be exactly n_envs '
                                 '({}) environments, but received {} '
                                 'environments.'.format(n, len(env_update)))
        else:
            env_update = [copy.deepcopy(env_update) for _ in range(n)]
        old = (list(self.env.envs) if isinstance(self.env, HostVecEnv)
               else [None] * n)
        members, updated = [], False
        for old_env, env_up in zip(old, env_update):
            env, up = _apply_env_update(old_env, env_up)
            members.append(env)
            updated = updated or up
        if not updated:
            return
        if any(m is None for m in members):
            raise TypeError('Unknown environment update type.')
        if self.env is not None and not isinstance(self.env, HostVecEnv):
            self.env.close()  # a device batch gives way to per-env objects
        # (the members of a previous HostVecEnv were kept, replaced or closed
        # one by one above: the batch object itself owns only staging buffers)
        self.env = HostVecEnv(members)
        self._needs_env_reset = True

    # -- rollout buffers ------------------------------------------------------
    def _alloc_buffers(self, num_samples):
        env, n = self.env, self._n_envs
        P = int(self._max_episode_length)
        # the last step t* satisfies t* <= ceil(num_samples / n) + P - 2
        tcap = -(-int(num_samples) // n) + P - 1
        tcap = round4(max(tcap, 1))
        if tcap > 65535 * 4:
            raise ValueError('rollout too long for one obtain_samples call')
        dev = self.device
        ldo = round4(env.obs_dim)
        lda = round4(env.act_width)
        ldh = round4(self.agent.net.out_dim)
        f32 = torch.float32
        b = {
            'Tcap': tcap,
            'obs': torch.empty(n, tcap, ldo, dtype=f32, device=dev),
            'act': torch.empty(n, tcap, lda, dtype=f32, device=dev),
            'head': (torch.empty(n, tcap, ldh, dtype=f32, device=dev)
                     if self._store_infos else None),
            'lastobs': torch.empty(n, tcap, ldo, dtype=f32, device=dev),
            'rew': torch.empty(n, tcap, dtype=f32, device=dev),
            'st': torch.empty(n, tcap, dtype=torch.uint8, device=dev),
            'tail': torch.zeros(n, tcap, dtype=torch.uint16, device=dev),
            'step_eps': torch.zeros(tcap, dtype=torch.int32, device=dev),
            'step_samples': torch.zeros(tcap, dtype=torch.int32, device=dev),
            'action': torch.zeros(n, lda, dtype=f32, device=dev),
            'done': torch.zeros(n, dtype=torch.uint8, device=dev),
            # per column: the n ``env_info`` dicts of a CPU env batch (or None)
            'infos': [None] * tcap,
            # column -> {env index: episode_info} for the episodes that ended
            # there (CPU env batches whose resets report any)
            'ep_infos': {},
        }
        if ldo != env.obs_dim:
            b['obs'].zero_()      # padding columns feed the GEMMs: keep them 0
            b['lastobs'].zero_()
        if lda != env.act_width:
            b['act'].zero_()
        return b

    def _start(self):
        """``vec_worker.py:107-137``: reset on agent / env update."""
        if not (self._needs_agent_reset or self._needs_env_reset):
            return
        self.agent.reset([True] * self._n_envs)
        if self._needs_env_reset or self._ep_t is None:
            self.env.reset_all()
            self._ep_t = torch.zeros(self._n_envs, dtype=torch.int32,
                                     device=self.device)
        else:
            # only environments with progress are reset (vec_worker.py:122-126)
            progress = (self._ep_t > 0).to(torch.uint8)
            self.env.hold()
            self.env.reset_where(progress)
            self.env.advance()
            self._ep_t.zero_()
        pop = getattr(self.env, 'pop_finished_episode_infos', None)
        if pop is not None:
            pop()  # episodes dropped by the reset are not part of any batch
        self._needs_agent_reset = False
        self._needs_env_reset = False

    def _step(self, b, col):
        """One vectorised step into column ``col`` of the rollout buffers."""
        env, pol, n = self.env, self.agent, self._n_envs
        fused = self._fused_ok()
        a = self._head_args(b, col, fused)
        s = stream_ptr()
        if fused:
            call('ga_policy_step_fused_f32', C.byref(pol.net._desc),
                 dptr(pol.net.params), C.byref(a), s)
        else:
            call('ga_policy_head_sample', C.byref(a), s)
        env.step_all(b['action'])
        b['infos'][col] = getattr(env, 'last_env_infos', None)
        r = self._record_args(b, col)
        call('ga_record_step', C.byref(r), s)
        env.reset_where(b['done'])
        pop = getattr(env, 'pop_finished_episode_infos', None)
        if pop is not None:
            finished = pop()
            if finished:
                b['ep_infos'][col] = dict(finished)
        env.advance()
        self._global_step += 1

    def _fused_ok(self):
        return self._use_fused and bool(
            _lib.load().ga_policy_step_fused_supported(
                C.byref(self.agent.net._desc)))

    def _head_args(self, b, col, fused):
        env, pol, n = self.env, self.agent, self._n_envs
        a = _lib.HeadArgs()
        a.n, a.env_id0 = n, getattr(env, 'env_id0', 0)
        a.kind = 0 if pol.kind == 'gaussian' else 1
        a.A = pol.net.out_dim
        if fused:
            a.ldh = round4(pol.net.out_dim)
        else:
            head = pol.net.forward(env.obs, n)
            a.head, a.ldh = head.data_ptr(), head.stride(0)
        if pol.kind == 'gaussian':
            a.log_std = pol.net.log_std.data_ptr()
            a.has_min, a.min_log_std, a.has_max, a.max_log_std = \
                pol._std_args()
        else:
            a.double_softmax = int(pol.double_softmax)
        noise = self._noise_fn(self._global_step) if self._noise_fn else None
        if noise is not None:
            self._noise_keepalive = noise
            a.noise, a.ldn = noise.data_ptr(), noise.stride(0)
        a.seed = (self._seed or 0) + 7919 * (self._worker_number + 1)
        a.step = self._global_step & 0xFFFFFFFF
        a.obs, a.ldo, a.obs_dim = (env.obs.data_ptr(), env.obs.stride(0),
                                   env.obs_dim)
        a.col, a.Tcap = col, b['Tcap']
        a.action, a.lda = b['action'].data_ptr(), b['action'].stride(0)
        a.obs_buf, a.act_buf = b['obs'].data_ptr(), b['act'].data_ptr()
        a.head_buf = b['head'].data_ptr() if b['head'] is not None else None
        return a

    def _record_args(self, b, col):
        env, n = self.env, self._n_envs
        r = _lib.RecordArgs()
        r.n, r.col, r.Tcap = n, col, b['Tcap']
        r.max_episode_length = int(self._max_episode_length)
        r.reward, r.step_type = env.reward.data_ptr(), env.step_type.data_ptr()
        r.next_obs, r.ldo, r.obs_dim = (env.next_obs.data_ptr(),
                                        env.next_obs.stride(0), env.obs_dim)
        r.ep_t = self._ep_t.data_ptr()
        r.rew_buf, r.st_buf = b['rew'].data_ptr(), b['st'].data_ptr()
        r.tail_buf, r.lastobs_buf = (b['tail'].data_ptr(),
                                     b['lastobs'].data_ptr())
        r.done = b['done'].data_ptr()
        r.step_eps = b['step_eps'].data_ptr()
        r.step_samples = b['step_samples'].data_ptr()
        r.terminal_only = getattr(self, '_terminal_only', 0)
        return r

    def _native_steps(self, b, col, n_steps):
        """``n_steps`` steps enqueued by ``ga_rollout_synth_steps`` (synthetic
        env, fused policy step, device RNG); False when not applicable."""
        from garage_amd.envs import NormalizedVecEnv, SyntheticVecEnv
        env = self.env
        inner, norm = env, None
        if type(env) is NormalizedVecEnv:  # statistics fused into the env step
            inner, norm = env._env, env.norm_args()
        if (n_steps <= 0 or type(inner) is not SyntheticVecEnv
                or self._noise_fn is not None or not self._fused_ok()):
            return False
        a = self._head_args(b, col, True)
        r = self._record_args(b, col)
        raw = norm is not None and norm.normalize_obs
        call('ga_rollout_synth_steps', C.byref(self.agent.net._desc),
             dptr(self.agent.net.params), C.byref(a), C.byref(inner._c),
             C.byref(r), dptr(env.obs), dptr(env.next_obs),
             None if norm is None else C.byref(norm),
             dptr(inner.obs) if raw else None,
             dptr(inner.next_obs) if raw else None, n_steps, stream_ptr())
        if n_steps % 2:
            env.advance()
        self._global_step += n_steps
        return True

    def rollout_samples(self, num_samples):
        """Everything ``LocalSampler.obtain_samples`` collects from one worker.

        Steps until the episodes completed so far hold ``>= num_samples``
        transitions, exactly like the ``while True: worker.rollout()`` loop of
        ``local_sampler.py:157-166`` (in-flight episodes at that point are
        dropped by the next call's reset).
        """
        self._start()
        n = self._n_envs
        # the reference loop runs at least one rollout(), i.e. until an episode
        # completes, whatever the target (local_sampler.py:157-166)
        num_samples = max(int(num_samples), 1)
        b = self._alloc_buffers(num_samples)
        col = 0
        # the target cannot be reached before step ceil(num_samples / n): those
        # steps need no host check and are enqueued natively when possible
        first_check = min(-(-int(num_samples) // n), b['Tcap'])
        if self._native_steps(b, 0, first_check):
            col = first_check
        while True:
            if col > 0 and col * n >= num_samples:  # cannot be reached any earlier
                done_samples = int(b['step_samples'][:col].sum().item())
                if done_samples >= num_samples:
                    break
            if col >= b['Tcap']:
                raise RuntimeError('rollout buffer exhausted: an environment '
                                   'ran past max_episode_length')
            self._step(b, col)
            col += 1
        return self._pack(b, col)

    def _pack(self, b, n_steps, first_step=0):
        """Episodes that ended in columns ``[first_step, n_steps)`` -> batch."""
        dev, n, tcap = self.device, self._n_envs, b['Tcap']
        step_eps = b['step_eps'][:n_steps].cpu().numpy().astype(np.int64)
        step_eps[:first_step] = 0
        ep_base = np.concatenate([[0], np.cumsum(step_eps)[:-1]])
        n_eps = int(step_eps.sum())
        s = stream_ptr()
        ep_env = torch.empty(n_eps, dtype=torch.int32, device=dev)
        ep_end = torch.empty_like(ep_env)
        ep_len = torch.empty_like(ep_env)
        base_dev = torch.from_numpy(ep_base.astype(np.int32)).to(dev)
        # columns before first_step are skipped by offsetting the column base
        tail = b['tail'][:, first_step:]
        call('ga_pack_episodes', dptr(tail), n, tcap, n_steps - first_step,
             dptr(base_dev[first_step:]), dptr(ep_env), dptr(ep_end),
             dptr(ep_len), s)
        if first_step:
            ep_end += first_step
        lengths = ep_len.cpu().numpy().astype(np.int64)
        off = np.concatenate([[0], np.cumsum(lengths)])
        S = int(off[-1])
        off_dev = torch.from_numpy(off).to(dev)
        src = torch.empty(S, dtype=torch.int32, device=dev)
        call('ga_pack_src_index', dptr(ep_env), dptr(ep_end), dptr(ep_len),
             dptr(off_dev), n_eps, tcap, dptr(src), s)
        ep_cell = (ep_env.long() * tcap + ep_end.long()).to(torch.int32)

        def rows(buf, idx, count):
            w = buf.shape[-1]
            out = torch.empty(count, w, dtype=torch.float32, device=dev)
            call('ga_gather_rows_f32', dptr(buf), w, dptr(idx), count, w,
                 dptr(out), w, s)
            return out

        obs = rows(b['obs'], src, S)
        act = rows(b['act'], src, S)
        head = rows(b['head'], src, S) if b['head'] is not None else None
        last = rows(b['lastobs'], ep_cell, n_eps)
        rew = torch.empty(S, dtype=torch.float32, device=dev)
        call('ga_gather_f32', dptr(b['rew']), dptr(src), S, dptr(rew), s)
        st = torch.empty(S, dtype=torch.uint8, device=dev)
        call('ga_gather_u8', dptr(b['st']), dptr(src), S, dptr(st), s)
        pol = self.agent
        gaussian = pol.kind == 'gaussian'
        return DeviceEpisodeBatch(
            self.env.spec, lengths=lengths, obs_dev=obs, last_obs_dev=last,
            actions_dev=act, rewards_dev=rew, step_types_dev=st,
            ep_off_dev=off_dev, head_dev=head,
            head_name='mean' if gaussian else 'prob',
            log_std=pol.clamped_log_std() if gaussian else None,
            discrete=is_discrete(self.env.spec.action_space),
            env_infos=self._packed_env_infos(b, src, tcap),
            episode_infos=self._packed_episode_infos(b, ep_env, ep_end))

    @staticmethod
    def _packed_episode_infos(b, ep_env, ep_end):
        """``episode_infos`` of the packed batch: per key an ``(N, ...)`` array,
        row ``e`` = what ``reset()`` reported for episode ``e``
        (``default_worker.py:94-96,158-161``; the reference's ``VecWorker`` loses
        them after an env's first episode, SURVEY.md Q23 -- the intended
        ``DefaultWorker`` layout is kept here)."""
        log = b.get('ep_infos')
        if not log or not any(info for col in log.values()
                              for info in col.values()):
            return {}
        envs = ep_env.cpu().numpy()
        ends = ep_end.cpu().numpy()
        rows = [log.get(int(c), {}).get(int(e), {})
                for e, c in zip(envs, ends)]
        keys = next(r for r in rows if r).keys()
        for e, r in enumerate(rows):
            missing = [k for k in keys if k not in r]
            if missing:
                # (the reference's EpisodeBatch.concatenate needs every episode
                # to carry the same episode_info keys, _dtypes.py:592-632)
                raise ValueError(
                    'episode {} (env {}, ended at step {}) reports no '
                    'episode_info {!r}; every reset() of a batch must report '
                    'the same keys'.format(e, int(envs[e]), int(ends[e]),
                                           missing))
        return {k: np.asarray([r[k] for r in rows]) for k in keys}

    @staticmethod
    def _packed_env_infos(b, src, tcap):
        """``env_infos`` of the packed batch: per key an ``(S, ...)`` array in the
        same sample order as the device arrays (``vec_worker.py:146-147,192-193``
        builds the same arrays per episode).  Only CPU env batches report any."""
        log = b.get('infos')
        if not log or all(entry is None for entry in log):
            return {}
        cells = src.cpu().numpy().astype(np.int64)  # env * tcap + column
        envs, cols = cells // tcap, cells % tcap
        first = log[int(cols[0])][int(envs[0])] if cells.size else {}
        out = {}
        for key in first:
            out[key] = np.asarray([log[int(c)][int(e)][key]
                                   for e, c in zip(envs, cols)])
        return out

    # -- reference-shaped API (sampler/worker.py:46-77) -------------------------
    # start_episode / step_episode / collect_episode run on one persistent
    # rollout buffer: a step is the same launches as everywhere else plus ONE
    # device sync (the count of episodes that ended in it), collect_episode packs
    # the columns stepped since the last collection.  rollout() is the three
    # calls in the reference's order (default_worker.py:176-186).
    def start_episode(self):
        """``vec_worker.py:107-137``."""
        self._start()

    def _api_buffer(self):
        if getattr(self, '_api_buf', None) is None:
            P = int(self._max_episode_length)
            self._api_buf = self._alloc_buffers(self._n_envs * P)
            self._api_col = 0    # next column to step into
            self._api_first = 0  # first column not yet handed out
        return self._api_buf

    def step_episode(self):
        """``vec_worker.py:176-204``: one vectorised step of every environment;
        True iff at least one episode completed in it."""
        if self._ep_t is None:
            self._start()
        b = self._api_buffer()
        if self._api_col >= b['Tcap']:
            b = self._api_buf = self._grown(b, self._api_col)
        col = self._api_col
        self._step(b, col)
        self._api_col = col + 1
        return int(b['step_eps'][col].item()) > 0

    def collect_episode(self):
        """``vec_worker.py:206-219``: the episodes completed since the last call,
        in (completion step, env) order."""
        b = self._api_buffer()
        first, col = self._api_first, self._api_col
        if col <= first or int(b['step_eps'][first:col].sum().item()) == 0:
            # (the reference falls into EpisodeBatch.concatenate() of nothing)
            raise ValueError('collect_episode(): no episode has completed '
                             'since the last call')
        batch = self._pack(b, col, first_step=first)
        self._api_first = col
        if col + int(self._max_episode_length) >= b['Tcap']:
            self._rebase_api_buffer()
        return batch

    def rollout(self):
        """``default_worker.py:176-186``: step until an episode completes."""
        self.start_episode()
        while not self.step_episode():
            pass
        return self.collect_episode()

    def _rebase_api_buffer(self):
        """Move the in-flight tails of the API buffer back to column 0 (every
        completed episode has been handed out: called by collect_episode)."""
        old, col = self._api_buf, self._api_col
        P = int(self._max_episode_length)
        new = self._alloc_buffers(self._n_envs * P)
        keep = min(col, P)
        for k in ('obs', 'act', 'head', 'lastobs', 'rew', 'st'):
            if old[k] is not None:
                new[k][:, :keep] = old[k][:, col - keep:col]
        new['infos'][:keep] = old['infos'][col - keep:col]
        new['ep_infos'] = {c - (col - keep): v
                           for c, v in old['ep_infos'].items()
                           if c >= col - keep}
        self._api_buf, self._api_col, self._api_first = new, keep, keep

    def _grown(self, old, col):
        """A rollout buffer of twice the columns holding columns ``[0, col)`` of
        ``old`` (steps taken without a collection in between)."""
        new = self._alloc_buffers(self._n_envs * 2 * old['Tcap'])
        for k in ('obs', 'act', 'head', 'lastobs', 'rew', 'st', 'tail'):
            if old[k] is not None:
                new[k][:, :col] = old[k][:, :col]
        for k in ('step_eps', 'step_samples'):
            new[k][:col] = old[k][:col]
        new['infos'][:col] = old['infos'][:col]
        new['ep_infos'] = dict(old['ep_infos'])
        return new

    def shutdown(self):
        if self.env is not None:
            self.env.close()

    def __getstate__(self):
        """Workers are not picklable (``sampler/worker.py:79-87``)."""
        raise ValueError('Workers are not pickleable. '
                         'Please pickle the WorkerFactory instead.')


class GpuFragmentWorker(GpuVecWorker):
    """``FragmentWorker`` (``sampler/fragment_worker.py:11-156``) on the GPU.

    Each ``rollout()`` advances every env by ``timesteps_per_call`` steps and
    returns *fragments*: the pieces of episodes collected during the call, cut
    at episode ends and at the end of the call.  Fragments that ended with their
    episode come first, in (step, env) order; the still-running ones follow in
    env order (``collect_episode``, ``:121-138``).  Episodes continue across
    calls (``update_agent`` does not reset them) and only TERMINAL -- not
    TIMEOUT -- or ``max_episode_length`` ends one (``:114-115``).
    """

    def __init__(self, *, seed, max_episode_length, worker_number,
                 n_envs=GpuVecWorker.DEFAULT_N_ENVS, timesteps_per_call=1,
                 noise_fn=None, store_agent_infos=True):
        super().__init__(seed=seed, max_episode_length=max_episode_length,
                         worker_number=worker_number, n_envs=n_envs,
                         noise_fn=noise_fn,
                         store_agent_infos=store_agent_infos)
        self._timesteps_per_call = int(timesteps_per_call)
        self._terminal_only = 1

    def update_agent(self, agent_update):
        """``default_worker.py:55-69``: no reset flag for fragment workers."""
        if isinstance(agent_update, (dict, tuple, np.ndarray)):
            self.agent.set_param_values(agent_update)
        elif agent_update is not None:
            self.agent = agent_update

    def _start(self):
        """``fragment_worker.py:89-96``: only an env update restarts episodes."""
        if self._needs_env_reset or self._ep_t is None:
            self.agent.reset([True] * self._n_envs)
            self.env.reset_all()
            self._ep_t = torch.zeros(self._n_envs, dtype=torch.int32,
                                     device=self.device)
        self._needs_agent_reset = False
        self._needs_env_reset = False

    def rollout_samples(self, num_samples):
        raise NotImplementedError('fragment workers are driven by rollout()')

    def start_episode(self):
        """``fragment_worker.py:85-91``."""
        self._start()

    def step_episode(self):
        """``fragment_worker.py:93-119``: one step of every environment; True
        iff an episode ended (TERMINAL or ``max_episode_length``) in it."""
        if self._ep_t is None:
            self._start()
        b = getattr(self, '_frag_buf', None)
        if b is None:
            b = self._frag_buf = self._alloc_buffers(
                self._n_envs * self._timesteps_per_call)
            self._frag_col = 0
        if self._frag_col >= b['Tcap']:
            b = self._frag_buf = self._grown(b, self._frag_col)
        col = self._frag_col
        self._step(b, col)
        self._frag_col = col + 1
        return int(b['step_eps'][col].item()) > 0

    def collect_episode(self):
        """``fragment_worker.py:121-138``: the fragments closed by an episode end
        since the last call in (step, env) order, then the still-running ones
        in env order."""
        b, tpc = getattr(self, '_frag_buf', None), getattr(self, '_frag_col', 0)
        if b is None or tpc == 0:
            raise ValueError('collect_episode(): no step has been taken since '
                             'the last call')
        n = self._n_envs
        self._frag_buf, self._frag_col = None, 0
        # fragment table on the host: the (n, tpc) tail matrix is tiny
        tail = b['tail'][:, :tpc].to(torch.int32).cpu().numpy()
        env_id, end, length = [], [], []
        cut = np.zeros(n, dtype=np.int64)  # first column of the open fragment
        for t in range(tpc):               # fragments closed by an episode end
            for i in np.nonzero(tail[:, t])[0]:
                env_id.append(i)
                end.append(t)
                length.append(t - cut[i] + 1)
                cut[i] = t + 1
        n_closed = len(env_id)
        for i in range(n):                 # still-running fragments, env order
            if cut[i] < tpc:
                env_id.append(i)
                end.append(tpc - 1)
                length.append(tpc - cut[i])
        return self._pack_fragments(b, np.asarray(env_id), np.asarray(end),
                                    np.asarray(length), n_closed)

    def rollout(self):
        """``fragment_worker.py:140-151``."""
        self.start_episode()
        for _ in range(self._timesteps_per_call):
            self.step_episode()
        return self.collect_episode()

    def _pack_fragments(self, b, env_id, end, length, n_closed):
        dev, tcap = self.device, b['Tcap']
        s = stream_ptr()
        n_frag = len(env_id)
        ep_env = torch.from_numpy(env_id.astype(np.int32)).to(dev)
        ep_end = torch.from_numpy(end.astype(np.int32)).to(dev)
        ep_len = torch.from_numpy(length.astype(np.int32)).to(dev)
        off = np.concatenate([[0], np.cumsum(length)]).astype(np.int64)
        S = int(off[-1])
        off_dev = torch.from_numpy(off).to(dev)
        src = torch.empty(S, dtype=torch.int32, device=dev)
        call('ga_pack_src_index', dptr(ep_env), dptr(ep_end), dptr(ep_len),
             dptr(off_dev), n_frag, tcap, dptr(src), s)

        def rows(buf, idx, count, ld=None):
            w = buf.shape[-1]
            out = torch.empty(count, w, dtype=torch.float32, device=dev)
            call('ga_gather_rows_f32', dptr(buf), w, dptr(idx), count, w,
                 dptr(out), w, s)
            return out

        obs = rows(b['obs'], src, S)
        act = rows(b['act'], src, S)
        head = rows(b['head'], src, S) if b['head'] is not None else None
        # last observation: the terminal one for closed fragments, the env's
        # current observation for the running ones
        last = torch.empty(n_frag, obs.shape[1], dtype=torch.float32,
                           device=dev)
        if n_closed:
            cell = (ep_env[:n_closed].long() * tcap +
                    ep_end[:n_closed].long()).to(torch.int32)
            last[:n_closed] = rows(b['lastobs'], cell, n_closed)
        if n_frag > n_closed:
            last[n_closed:] = rows(self.env.obs, ep_env[n_closed:].contiguous(),
                                   n_frag - n_closed)
        rew = torch.empty(S, dtype=torch.float32, device=dev)
        call('ga_gather_f32', dptr(b['rew']), dptr(src), S, dptr(rew), s)
        st = torch.empty(S, dtype=torch.uint8, device=dev)
        call('ga_gather_u8', dptr(b['st']), dptr(src), S, dptr(st), s)
        pol = self.agent
        gaussian = pol.kind == 'gaussian'
        return DeviceEpisodeBatch(
            self.env.spec, lengths=length.astype(np.int64), obs_dev=obs,
            last_obs_dev=last, actions_dev=act, rewards_dev=rew,
            step_types_dev=st, ep_off_dev=off_dev, head_dev=head,
            head_name='mean' if gaussian else 'prob',
            log_std=pol.clamped_log_std() if gaussian else None,
            discrete=is_discrete(self.env.spec.action_space),
            env_infos=self._packed_env_infos(b, src, tcap),
            episode_infos=self._packed_episode_infos(b, ep_env, ep_end))


class GpuVecSampler:
    """``LocalSampler`` (``sampler/local_sampler.py:13-232``) for GPU workers.

    Same constructor keywords, ``from_worker_factory``, ``obtain_samples``,
    ``obtain_exact_episodes``, ``shutdown_worker``, ``total_env_steps`` and
    pickling behaviour; ``envs`` may be a :class:`~garage_amd.envs.VecEnv`.
    Returns :class:`~garage_amd._dtypes.DeviceEpisodeBatch`.
    """

    def __init__(self, agents, envs, *, worker_factory=None,
                 max_episode_length=None, is_tf_worker=False, seed=None,
                 n_workers=1, worker_class=GpuVecWorker, worker_args=None):
        if worker_factory is None and max_episode_length is None:
            raise TypeError('Must construct a sampler from WorkerFactory or'
                            'parameters (at least max_episode_length)')
        if isinstance(worker_factory, WorkerFactory):
            self._factory = worker_factory
        else:
            self._factory = WorkerFactory(
                max_episode_length=max_episode_length,
                is_tf_worker=is_tf_worker, seed=seed, n_workers=n_workers,
                worker_class=worker_class, worker_args=worker_args)
        self._agents = self._factory.prepare_worker_messages(agents)
        self._envs = self._factory.prepare_worker_messages(
            envs, preprocess=_copy_env)
        self._build_workers()
        self.total_env_steps = 0

    def _build_workers(self):
        self._workers = [
            self._factory(i) for i in range(self._factory.n_workers)
        ]
        for worker, agent, env in zip(self._workers, self._agents, self._envs):
            worker.update_agent(agent)
            worker.update_env(env)

    @classmethod
    def from_worker_factory(cls, worker_factory, agents, envs):
        return cls(agents, envs, worker_factory=worker_factory)

    def start_worker(self):
        """No-op, as in ``sampler/sampler.py``."""

    def _update_workers(self, agent_update, env_update):
        agent_updates = self._factory.prepare_worker_messages(agent_update)
        env_updates = self._factory.prepare_worker_messages(
            env_update, preprocess=_copy_env)
        for worker, a, e in zip(self._workers, agent_updates, env_updates):
            worker.update_agent(a)
            worker.update_env(e)

    def obtain_samples(self, itr, num_samples, agent_update, env_update=None):
        """``local_sampler.py:134-166``."""
        del itr
        self._update_workers(agent_update, env_update)
        if len(self._workers) == 1 and not isinstance(self._workers[0],
                                                      GpuFragmentWorker):
            samples = self._workers[0].rollout_samples(num_samples)
        else:
            batches, done = [], 0
            while done < num_samples:
                for worker in self._workers:
                    batch = worker.rollout()
                    done += len(batch.actions)
                    batches.append(batch.to_host())
                    if done >= num_samples:
                        break
            samples = EpisodeBatch.concatenate(*batches)
        self.total_env_steps += int(sum(samples.lengths))
        return samples

    def obtain_exact_episodes(self, n_eps_per_worker, agent_update,
                              env_update=None):
        """``local_sampler.py:168-200`` (worker order)."""
        self._update_workers(agent_update, env_update)
        batches = []
        for worker in self._workers:
            for _ in range(n_eps_per_worker):
                batches.append(worker.rollout().to_host())
        samples = EpisodeBatch.concatenate(*batches)
        self.total_env_steps += int(sum(samples.lengths))
        return samples

    def shutdown_worker(self):
        for worker in self._workers:
            worker.shutdown()

    def __getstate__(self):
        state = self.__dict__.copy()
        state['_workers'] = None  # local_sampler.py:207-217
        # what a worker carries from one rollout to the next: the Philox step
        # counter of the action noise and which envs have an episode in flight
        # (only those are reset at the next rollout, vec_worker.py:122-126), so
        # that a resumed run (trainer.py:263-341) continues like the
        # uninterrupted one
        state['_worker_state'] = [
            dict(global_step=getattr(w, '_global_step', 0),
                 ep_t=None if getattr(w, '_ep_t', None) is None else
                 w._ep_t.cpu().numpy())
            for w in self._workers]
        return state

    def __setstate__(self, state):
        saved = state.pop('_worker_state', None)
        self.__dict__.update(state)
        self._build_workers()
        for w, st in zip(self._workers, saved or []):
            w._global_step = st['global_step']
            if st['ep_t'] is not None:
                w._ep_t = torch.from_numpy(st['ep_t']).to(w.device)
                w._needs_env_reset = False  # the env came back with its state


__all__ = ['WorkerFactory', 'GpuVecWorker', 'GpuFragmentWorker',
           'GpuVecSampler']
