"""Host-side contract of the drop-in surface (no GPU): EpisodeBatch validation
and views, step types, factory / sampler / algorithm argument errors.  Each
case mirrors a reference test (file:line in the docstrings)."""
import numpy as np
import pytest

from garage_amd._dtypes import (Box, EnvSpec, EpisodeBatch, StepType,
                                pad_batch_array)


@pytest.fixture
def eps_data():
    """tests/garage/test_dtypes.py:14-63 (a Box stands in for MultiDiscrete)."""
    obs_space = Box(low=1, high=np.inf, shape=(4, 3, 2), dtype=np.float32)
    act_space = Box(low=0, high=5, shape=(2, ), dtype=np.float32)
    env_spec = EnvSpec(obs_space, act_space, max_episode_length=100)
    lens = np.array([10, 20, 7, 25, 25, 40, 10, 5])
    n_t = lens.sum()
    obs = np.stack([obs_space.low] * n_t)
    last_obs = np.stack([obs_space.low] * len(lens))
    act = np.stack([[1., 3.]] * n_t).astype(np.float32)
    step_types = []
    for size in lens:
        step_types.extend([StepType.FIRST] + [StepType.MID] * (size - 2) +
                          [StepType.TERMINAL])
    return {
        'episode_infos': {'task_one_hot': np.stack([[1, 1]] * len(lens))},
        'env_spec': env_spec,
        'observations': obs,
        'last_observations': last_obs,
        'actions': act,
        'rewards': np.arange(n_t),
        'env_infos': {'goal': np.stack([[1, 1]] * n_t), 'foo': np.arange(n_t)},
        'agent_infos': {'prev_action': act, 'hidden': np.arange(n_t)},
        'step_types': np.array(step_types, dtype=StepType),
        'lengths': lens,
    }


def test_new_eps(eps_data):
    """tests/garage/test_dtypes.py:66-79."""
    t = EpisodeBatch(**eps_data)
    for k in ('env_spec', 'observations', 'last_observations', 'actions',
              'rewards', 'env_infos', 'agent_infos', 'step_types', 'lengths'):
        assert getattr(t, k) is eps_data[k]
    assert t.episode_infos_by_episode is eps_data['episode_infos']
    assert t.episode_infos['task_one_hot'].shape == (eps_data['lengths'].sum(),
                                                     2)


@pytest.mark.parametrize('mutate,match', [
    (lambda d: d.update(lengths=d['lengths'].reshape((4, -1))),
     'lengths has shape'),
    (lambda d: d.update(lengths=d['lengths'].astype(np.float32)),
     'lengths has dtype float32'),
    (lambda d: d.update(observations=d['observations'][:, :, :, :1]),
     'Each observation has shape'),
    (lambda d: d.update(observations=d['observations'][:-1]),
     'observations has batch size'),
    (lambda d: d.update(last_observations=d['last_observations'][:, :, :, :1]),
     'last_observations must have the '),
    (lambda d: d.update(last_observations=d['last_observations'][:-1]),
     'last_observations has batch size 7'),
    (lambda d: d.update(actions=d['actions'][:, 0]), 'Each action has shape '),
    (lambda d: d.update(actions=d['actions'][:-1]), 'actions has batch size'),
    (lambda d: d.update(rewards=d['rewards'].reshape((2, -1))),
     'rewards has shape'),
    (lambda d: d['env_infos'].update(bar=[]), "Entry 'bar' in env_infos"),
    (lambda d: d['env_infos'].update(goal=d['env_infos']['goal'][:-1]),
     "Entry 'goal' in env_infos has batch size 141"),
    (lambda d: d['agent_infos'].update(bar=list()),
     "Entry 'bar' in agent_infos"),
    (lambda d: d['agent_infos'].update(hidden=d['agent_infos']['hidden'][:-1]),
     "Entry 'hidden' in agent_infos has batch size 141"),
    (lambda d: d.update(step_types=d['step_types'].reshape((2, -1))),
     'step_types has batch size 2'),
    (lambda d: d.update(step_types=d['step_types'].astype(np.float32)),
     'step_types has dtype float32'),
    (lambda d: d['episode_infos'].update(bar=list()),
     "Entry 'bar' in episode_infos"),
    (lambda d: d['episode_infos'].update(
        task_one_hot=d['episode_infos']['task_one_hot'][:-1]),
     "Entry 'task_one_hot' in episode_infos"),
])
def test_eps_validation_errors(eps_data, mutate, match):
    """tests/garage/test_dtypes.py:82-216: same ValueError messages."""
    mutate(eps_data)
    with pytest.raises(ValueError, match=match):
        EpisodeBatch(**eps_data)


def test_padded_views_and_valids(eps_data):
    """tests/garage/test_dtypes.py:238-273."""
    t = EpisodeBatch(**eps_data)
    lens = eps_data['lengths']
    assert t.padded_observations.shape == (8, 100, 4, 3, 2)
    assert t.padded_rewards.shape == (8, 100)
    assert t.valids.shape == (8, 100)
    assert np.array_equal(t.valids.sum(axis=1), lens)
    start = 0
    for i, n in enumerate(lens):
        assert np.array_equal(t.padded_rewards[i, :n],
                              t.rewards[start:start + n])
        assert not t.padded_rewards[i, n:].any()
        start += n


def test_split_concatenate_roundtrip(eps_data):
    """tests/garage/test_dtypes.py:219-235 + _dtypes.py:592-674."""
    t = EpisodeBatch(**eps_data)
    parts = t.split()
    assert [int(p.lengths[0]) for p in parts] == list(eps_data['lengths'])
    back = EpisodeBatch.concatenate(*parts)
    assert np.array_equal(back.observations, t.observations)
    assert np.array_equal(back.rewards, t.rewards)
    assert np.array_equal(back.lengths, t.lengths)
    assert np.array_equal(back.env_infos['foo'], t.env_infos['foo'])
    nxt = t.next_observations
    assert nxt.shape == t.observations.shape


def test_pad_batch_array():
    """tests/garage/np/test_functions.py:81-88."""
    out = pad_batch_array(np.arange(10), [1, 2, 3, 4])
    assert out.shape == (4, 4)
    assert (out[2] == [3, 4, 5, 0]).all()
    with pytest.warns(UserWarning):
        wide = pad_batch_array(np.arange(10), [1, 2, 3, 4], max_length=2)
    assert wide.shape == (4, 4)


def test_step_type_table():
    """tests/garage/test_dtypes.py:290-318."""
    g = StepType.get_step_type
    assert g(step_cnt=1, max_episode_length=5, done=False) == StepType.FIRST
    assert g(step_cnt=2, max_episode_length=5, done=False) == StepType.MID
    assert g(step_cnt=2, max_episode_length=None, done=False) == StepType.MID
    assert g(step_cnt=5, max_episode_length=5, done=False) == StepType.TIMEOUT
    assert g(step_cnt=5, max_episode_length=5, done=True) == StepType.TIMEOUT
    assert g(step_cnt=1, max_episode_length=5, done=True) == StepType.TERMINAL
    with pytest.raises(ValueError):
        g(step_cnt=0, max_episode_length=5, done=False)


def test_sampler_and_factory_argument_errors():
    """tests/garage/sampler/test_local_sampler.py:118-126 and
    sampler/worker_factory.py:89-92,107-108."""
    from garage_amd.sampler import GpuVecSampler, WorkerFactory
    with pytest.raises(TypeError, match='Must construct a sampler from'):
        GpuVecSampler(agents=None, envs=None)
    wf = WorkerFactory(max_episode_length=5, n_workers=2)
    with pytest.raises(ValueError, match="Length of list doesn't match"):
        wf.prepare_worker_messages([1, 2, 3])
    assert wf.prepare_worker_messages('x') == ['x', 'x']
    with pytest.raises(ValueError, match='Worker number is too big'):
        wf(2)


def test_entropy_configuration_errors():
    """tests/garage/torch/algos/test_vpg.py:16-29,100-106."""
    from garage_amd.algos import VPG
    check = VPG._check_entropy_configuration
    with pytest.raises(ValueError, match='Invalid entropy_method'):
        check('bogus', False, True, 0.1)
    with pytest.raises(ValueError, match='center_adv should be False'):
        check('max', True, True, 0.1)
    with pytest.raises(ValueError, match='stop_gradient should be True'):
        check('max', False, False, 0.1)
    with pytest.raises(ValueError, match='policy_ent_coeff should be zero'):
        check('no_entropy', True, False, 0.1)
    check('regularized', True, False, 0.1)
    check('no_entropy', True, False, 0.0)


def test_optimizer_wrapper_arguments():
    """_functions.py:25-65 make_optimizer forms; the default-shaped Adam is the
    fused one (other torch.optim classes: the test below)."""
    import torch

    from garage_amd.optimizers import _parse_optimizer
    assert _parse_optimizer(torch.optim.Adam)['lr'] == 1e-3
    h = _parse_optimizer((torch.optim.Adam, dict(lr=2.5e-4, eps=1e-5)))
    assert h['lr'] == 2.5e-4 and h['eps'] == 1e-5 and h['kind'] == 'adam'
    with pytest.raises(NotImplementedError):
        _parse_optimizer(torch.optim.LBFGS)


def test_optimizer_parsing_matches_torch_defaults():
    """``make_optimizer`` accepts a type or ``(type, kwargs)`` (``_functions.py:
    25-65``): the settings handed to the kernels carry torch's own defaults, the
    default-shaped Adam is the fused one, unknown classes / options are refused."""
    import torch

    from garage_amd.optimizers import _parse_optimizer
    h = _parse_optimizer(torch.optim.Adam)
    assert h == dict(kind='adam', lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    h = _parse_optimizer((torch.optim.Adam, dict(lr=3e-4, amsgrad=True)))
    assert h['kind'] == 'generic' and h['code'] == 3 and h['flags'] == 1
    assert h['h'] == [3e-4, 0.9, 0.999, 1e-8, 0] and h['needs'] == (True, True, True)
    h = _parse_optimizer(torch.optim.AdamW)
    assert h['flags'] == 2 and h['h'][4] == 1e-2  # torch's AdamW default decay
    h = _parse_optimizer((torch.optim.SGD, dict(lr=0.1, momentum=0.9)))
    assert h['code'] == 1 and h['h'] == [0.1, 0.9, 0, 0, 0.0]
    assert h['needs'] == (True, False, False)
    h = _parse_optimizer(torch.optim.RMSprop)
    assert h['code'] == 2 and h['h'] == [1e-2, 0.99, 1e-8, 0, 0]
    with pytest.raises(ValueError, match='Nesterov'):
        _parse_optimizer((torch.optim.SGD, dict(nesterov=True)))
    with pytest.raises(NotImplementedError):
        _parse_optimizer(torch.optim.Adagrad)
    with pytest.raises(NotImplementedError):
        _parse_optimizer((torch.optim.SGD, dict(maximize=True)))
    with pytest.raises(NotImplementedError):
        _parse_optimizer((torch.optim.Adam, dict(bogus=1)))


def test_numpy_minibatch_stream_equals_batchdataset_semantics():
    """np/optimizers/minibatch_dataset.py:4-35 / optimizer_wrapper.py:31-49:
    one shuffle at construction + one per pass, cumulative, global numpy RNG;
    checked against the oracle's restatement (itself pinned by the real PPO
    goldens, whose parameters depend on every permutation)."""
    from oracle.batch import minibatch_index_stream
    n, mb, E = 23, 5, 3
    np.random.seed(4)
    want = minibatch_index_stream(n, mb, E)
    # the product draws the same ids; device upload is the only GPU part, so
    # restate its host half here
    np.random.seed(4)
    ids = np.arange(n, dtype=np.int32)
    np.random.shuffle(ids)
    got = []
    for _ in range(E):
        perm = ids.copy()
        got += [perm[k * mb:(k + 1) * mb] for k in range(-(-n // mb))]
        np.random.shuffle(ids)
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)


def test_data_parallel_ranks_take_the_same_number_of_optimizer_steps():
    """one gradient all-reduce per optimizer step, so ranks
    with different sample counts must still run the same number of minibatches
    (ceil(n / ceil(n / K)) != K in general: 9900 and 9921 samples at the
    reference-default minibatch of 64 gave 155 and 156 steps before)."""
    import torch

    from garage_amd.optimizers import OptimizerWrapper, data_parallel_plan
    rng = np.random.RandomState(0)
    pairs = [(9900, 9921), (10000, 10000), (64, 1), (65, 130)]
    pairs += [tuple(rng.randint(9000, 11000, size=2)) for _ in range(200)]
    pairs += [tuple(rng.randint(100, 400, size=3)) for _ in range(50)]
    for counts in pairs:
        if min(counts) < -(-max(counts) // 64):
            with pytest.raises(RuntimeError, match='fewer samples'):
                data_parallel_plan(counts, 64, 0)
            continue
        plans = [data_parallel_plan(counts, 64, r) for r in range(len(counts))]
        K = plans[0][0]
        assert all(p[0] == K for p in plans)
        total = np.sum([p[1].astype(np.float64) for p in plans], axis=0)
        assert np.allclose(total, 1.0, atol=1e-6)
        sizes = []
        for r, n in enumerate(counts):
            opt = OptimizerWrapper(torch.optim.Adam, None, 2, 64)
            opt.dp_minibatches, opt.dp_grad_scales = plans[r]
            b = opt.minibatch_bounds(n)
            assert len(b) == K + 1 and b[0] == 0 and b[-1] == n
            assert all(b[k + 1] > b[k] for k in range(K))
            assert opt.local_minibatch_size(n) == max(np.diff(b))
            sizes.append(np.diff(b))
        sizes = np.asarray(sizes, dtype=np.float64)
        for r in range(len(counts)):
            assert np.allclose(plans[r][1], sizes[r] / sizes.sum(axis=0),
                               atol=1e-7)
    # single process: BatchDataset's minibatches, the last one partial
    opt = OptimizerWrapper(torch.optim.Adam, None, 1, 5)
    assert opt.minibatch_bounds(23) == [0, 5, 10, 15, 20, 23]
    assert opt.minibatch_bounds(20) == [0, 5, 10, 15, 20]
    assert OptimizerWrapper(torch.optim.Adam, None).minibatch_bounds(7) == [0, 7]


def test_native_minibatch_split_equals_python_bounds():
    """``ga_update_epoch*`` (C++ ``minibatch_range``) and
    ``OptimizerWrapper.minibatch_bounds`` must cut a pass at the same ids on every
    rank -- workspaces and the per-step gradient weights of data-parallel runs
    are sized from the Python side (round-2 advisor finding).  Ragged counts,
    both split modes (``BatchDataset``'s ``mb``-sized minibatches,
    ``np/optimizers/minibatch_dataset.py:20-35``; the even split into
    ``dp_minibatches``), through the host-only C entry point."""
    import ctypes as C

    import torch

    from garage_amd import _lib
    from garage_amd.optimizers import OptimizerWrapper, data_parallel_plan
    lib = _lib.load()

    def native(S, mb, n_mb, has_perm):
        start, M = C.c_int64(), C.c_int64()
        n = lib.ga_minibatch_range(S, mb, n_mb, has_perm, 0, None, None)
        assert n >= 1, _lib.last_error() if hasattr(_lib, 'last_error') else n
        b = []
        for k in range(n):
            assert lib.ga_minibatch_range(S, mb, n_mb, has_perm, k,
                                          C.byref(start), C.byref(M)) == n
            b.append((start.value, M.value))
        assert all(b[k][0] + b[k][1] == b[k + 1][0] for k in range(n - 1))
        return [x[0] for x in b] + [b[-1][0] + b[-1][1]]

    rng = np.random.RandomState(1)
    for S in [1, 5, 63, 64, 65, 127, 128, 1000, 9900, 9921, 16384, 1048576] + \
            [int(v) for v in rng.randint(1, 20000, size=100)]:
        for mb in (1, 7, 64, 4096, 32768):
            opt = OptimizerWrapper(torch.optim.Adam, None, 1, mb)
            assert native(S, mb, 0, 1) == opt.minibatch_bounds(S), (S, mb)
        # no minibatching: one full batch, no permutation
        assert native(S, 0, 0, 0) == \
            OptimizerWrapper(torch.optim.Adam, None).minibatch_bounds(S)
    for _ in range(200):
        counts = tuple(int(v) for v in rng.randint(300, 12000,
                                                   size=rng.randint(2, 9)))
        if min(counts) < -(-max(counts) // 64):
            continue
        for r, n in enumerate(counts):
            opt = OptimizerWrapper(torch.optim.Adam, None, 1, 64)
            opt.dp_minibatches, opt.dp_grad_scales = \
                data_parallel_plan(counts, 64, r)
            assert native(n, 64, int(opt.dp_minibatches), 1) == \
                opt.minibatch_bounds(n), (counts, r)
    assert lib.ga_minibatch_range(0, 64, 0, 1, 0, None, None) < 0
    assert lib.ga_minibatch_range(10, 64, 0, 1, 1, None, None) < 0


def test_step_types_as_uint8_fast_and_fallback_paths():
    """Object arrays of the enum singletons (what garage's EpisodeBatch holds),
    integer arrays, mixed objects (plain ints among the members) and empties."""
    from garage_amd._dtypes import StepType, step_types_as_uint8
    rng = np.random.RandomState(0)
    codes = rng.randint(0, 4, size=1000).astype(np.uint8)
    members = [StepType(int(c)) for c in codes]
    out = step_types_as_uint8(np.array(members, dtype=object))
    assert out.dtype == np.uint8 and np.array_equal(out, codes)
    assert np.array_equal(step_types_as_uint8(codes.astype(np.int64)), codes)
    mixed = np.array(members, dtype=object)
    mixed[::7] = [int(c) for c in codes[::7]]  # plain ints: not the singletons
    assert np.array_equal(step_types_as_uint8(mixed), codes)
    assert step_types_as_uint8(np.array([], dtype=object)).shape == (0, )
    strided = np.array(members, dtype=object)[::3]  # non-contiguous view
    assert np.array_equal(step_types_as_uint8(strided), codes[::3])


def test_discrete_space_flatten_is_one_hot():
    """``akro.Discrete.flatten`` / ``flatten_n`` / ``unflatten`` semantics."""
    from garage_amd._dtypes import Discrete
    sp = Discrete(5)
    assert sp.flat_dim == 5 and sp.contains(np.int64(4)) and not sp.contains(5)
    assert np.array_equal(sp.flatten(3), [0, 0, 0, 1, 0])
    assert np.array_equal(sp.flatten_n([0, 4]),
                          [[1, 0, 0, 0, 0], [0, 0, 0, 0, 1]])
    assert sp.unflatten(sp.flatten(2)) == 2


def test_episode_batch_accessors_match_the_real_reference(golden):
    """tests/golden/episode_batch_methods.npz (the real ``garage.EpisodeBatch``,
    ``_dtypes.py:381-390,676-977``): per-episode lists, padded infos and next
    observations, terminals, ``to_list`` -- including its quirk of slicing the
    per-step expansion of ``episode_infos`` by the episode index -- and
    ``from_list`` for paths with T + 1 observations, with ``next_observations``
    and with neither, with ``dones`` standing in for step types."""
    g = golden('episode_batch_methods')
    P = int(g['P'])
    spec = EnvSpec(Box(-np.inf, np.inf, (3, )), Box(-np.inf, np.inf, (2, )),
                   max_episode_length=P)
    st = np.asarray([StepType(int(s)) for s in g['in_step_types']],
                    dtype=StepType)
    eps = EpisodeBatch(
        env_spec=spec,
        episode_infos={'goal': g['in_ep_goal'], 'task': g['in_ep_task']},
        observations=g['in_observations'],
        last_observations=g['in_last_observations'], actions=g['in_actions'],
        rewards=g['in_rewards'],
        env_infos={'success': g['in_env_success'], 'pos': g['in_env_pos']},
        agent_infos={'mean': g['in_agent_mean']}, step_types=st,
        lengths=g['in_lengths'])
    assert np.array_equal(eps.terminals, g['terminals'])
    assert np.array_equal(eps.padded_next_observations,
                          g['padded_next_observations'])
    assert np.array_equal(eps.padded_actions, g['padded_actions'])
    assert np.array_equal(
        np.asarray([[int(s) for s in row] for row in eps.padded_step_types]),
        g['padded_step_types'])
    assert np.array_equal(eps.padded_agent_infos['mean'],
                          g['padded_agent_mean'])
    assert np.array_equal(eps.padded_env_infos['pos'], g['padded_env_pos'])
    assert np.array_equal(eps.padded_env_infos['success'],
                          g['padded_env_success'])
    assert np.array_equal(eps.next_observations, g['next_observations'])
    assert np.array_equal(eps.episode_infos['goal'], g['episode_infos_goal'])
    n = len(eps.lengths)
    for i, (o, a) in enumerate(zip(eps.observations_list, eps.actions_list)):
        assert np.array_equal(o, g['list%d_obs' % i])
        assert np.array_equal(a, g['list%d_act' % i])
    dicts = eps.to_list()
    assert len(dicts) == n
    for i, d in enumerate(dicts):
        for k in ('observations', 'next_observations', 'actions', 'rewards'):
            assert np.array_equal(d[k], g['tolist%d_%s' % (i, k)]), (i, k)
        assert np.array_equal([int(s) for s in d['step_types']],
                              g['tolist%d_step_types' % i])
        assert np.array_equal(d['episode_infos']['goal'],
                              g['tolist%d_ep_goal' % i])
        assert np.array_equal(d['env_infos']['pos'], g['tolist%d_env_pos' % i])
        assert np.array_equal(d['agent_infos']['mean'],
                              g['tolist%d_agent_mean' % i])
    paths = []
    for i, d in enumerate(dicts):
        paths.append(dict(
            episode_infos={'goal': g['in_ep_goal'][i]},
            observations=np.concatenate([d['observations'],
                                         d['next_observations'][-1:]]),
            actions=d['actions'], rewards=d['rewards'],
            env_infos=d['env_infos'], agent_infos=d['agent_infos'],
            dones=np.asarray([int(s) == 2 for s in d['step_types']])))

    def same(batch, prefix):
        for k in ('observations', 'last_observations', 'actions', 'rewards',
                  'lengths'):
            assert np.array_equal(getattr(batch, k), g[prefix + k]), k
        assert np.array_equal([int(s) for s in batch.step_types],
                              g[prefix + 'step_types'])
        assert str(batch.lengths.dtype) == str(g[prefix + 'lengths_dtype'])

    back = EpisodeBatch.from_list(spec, paths)
    same(back, 'tp1_')
    assert np.array_equal(back.episode_infos_by_episode['goal'],
                          g['tp1_ep_goal'])
    paths2 = [dict(p, observations=p['observations'][:-1],
                   next_observations=d['next_observations'])
              for p, d in zip(paths, dicts)]
    same(EpisodeBatch.from_list(spec, paths2), 'nxt_')
    paths3 = [dict(p, observations=p['observations'][:-1]) for p in paths]
    assert np.array_equal(
        EpisodeBatch.from_list(spec, paths3).last_observations,
        g['bare_last_observations'])
