"""Data-parallel PPO on real kernels: 2 ranks (gloo, sharing the one GPU of the
test box) == the single-process oracle on the concatenated batch.

Full-batch optimizer steps (``minibatch_size=None``) make the comparison exact:
the global gradient is the mean over the union of the shards, i.e. the average
of the rank means for equal shard sizes, and ``center_adv`` must use the global
moments.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

N_EPS, P, O, A = 12, 10, 5, 3  # per rank


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _make_shard(rank, n_eps=N_EPS):
    rng = np.random.RandomState(100 + rank)
    N_EPS = n_eps
    lens = rng.randint(3, P + 1, size=N_EPS)
    lens[0] = P
    S = int(lens.sum())
    st = []
    for L in lens:
        t = [1] * L
        t[0] = 0
        t[-1] = 3 if L == P else 2
        st += t
    return dict(observations=rng.randn(S, O).astype(np.float32),
                last_observations=rng.randn(N_EPS, O).astype(np.float32),
                actions=rng.randn(S, A).astype(np.float32),
                rewards=rng.randn(S), step_types=np.asarray(st), lengths=lens)


def _rank_main(rank, world, port, q, init_pol, init_vf, algo_name='ppo',
               backend='gloo', minibatch=None, n_eps=None, hidden=(16, 16)):
    try:
        os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank),
                          WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                          MASTER_PORT=str(port), GARAGE_AMD_BACKEND=backend)
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        from garage_amd._dtypes import Box, EnvSpec, EpisodeBatch, StepType
        from garage_amd.algos import PPO
        from garage_amd.distributed import init_from_env, shard_algo
        from garage_amd.optimizers import OptimizerWrapper
        from garage_amd.policies import (GaussianMLPPolicy,
                                         GaussianMLPValueFunction)
        comm = init_from_env()
        if comm is None:  # a world of one rank: still a real process group
            import torch.distributed as dist
            from garage_amd.distributed import Comm
            torch.cuda.set_device(0)
            dist.init_process_group(backend=backend, rank=0, world_size=1)
            comm = Comm()
        spec = EnvSpec(Box(-np.inf, np.inf, (O, )), Box(-np.inf, np.inf, (A, )),
                       max_episode_length=P)
        pol = GaussianMLPPolicy(spec, hidden_sizes=hidden)
        vf = GaussianMLPValueFunction(spec, hidden_sizes=hidden)
        if rank == 0:  # rank 0's parameters must win (broadcast in shard_algo)
            pol.load_state_dict(init_pol)
            vf.load_state_dict(init_vf)
        opt = (torch.optim.Adam, dict(lr=1e-3))
        if algo_name == 'trpo':
            from garage_amd.algos import TRPO
            from garage_amd.optimizers import ConjugateGradientOptimizer
            algo = TRPO(env_spec=spec, policy=pol, value_function=vf,
                        sampler=None,
                        policy_optimizer=OptimizerWrapper(
                            (ConjugateGradientOptimizer,
                             dict(max_constraint_value=0.01)), pol),
                        vf_optimizer=OptimizerWrapper(opt, vf, 3, None))
        else:
            algo = PPO(env_spec=spec, policy=pol, value_function=vf,
                       sampler=None,
                       policy_optimizer=OptimizerWrapper(opt, pol, 3,
                                                         minibatch),
                       vf_optimizer=OptimizerWrapper(opt, vf, 3, minibatch))
        shard_algo(algo, comm)
        d = _make_shard(rank, *([] if n_eps is None else [n_eps[rank]]))
        np.random.seed(50 + rank)  # the rank's host permutation stream
        st = np.asarray([StepType(int(s)) for s in d['step_types']],
                        dtype=object)
        batch = EpisodeBatch(env_spec=spec, episode_infos={},
                             observations=d['observations'],
                             last_observations=d['last_observations'],
                             actions=d['actions'], rewards=d['rewards'],
                             env_infos={}, agent_infos={}, step_types=st,
                             lengths=d['lengths'])
        algo._train_once(0, batch)
        out = {k: v.numpy() for k, v in pol.state_dict().items()}
        out.update({'vf:' + k: v.numpy() for k, v in vf.state_dict().items()})
        out['tab'] = dict(algo.last_tabular)
        out['native_comm'] = algo._policy_optimizer.native_comm is not None \
            and algo._vf_optimizer.native_comm is not None
        out['native_loop'] = bool(algo._native_update_ok())
        out['adam_steps'] = (pol.net.adam_steps, vf.net.adam_steps)
        if algo_name == 'trpo':
            out['accepted'] = algo.last_cg['accepted']
        import torch.distributed as dist
        dist.destroy_process_group()
        q.put((rank, 'ok', out))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, 'error: %r\n%s' % (e, traceback.format_exc()), None))


@pytest.mark.timeout(300)
def test_two_rank_ppo_equals_single_process_oracle():
    from oracle import batch as ob
    from oracle import networks as nets
    from oracle.ppo import OraclePPO
    rng = np.random.RandomState(0)
    init_pol = nets.init_gaussian_mlp(rng, nets.POLICY_PREFIX, O, A, (16, 16),
                                      min_std=1e-6)
    init_vf = nets.init_gaussian_mlp(rng, nets.VALUE_PREFIX, O, 1, (16, 16))
    for p in (init_pol, init_vf):  # biases / log-std away from their inits
        for k in p:
            if 'min_std' not in k:
                p[k] = p[k] + torch.from_numpy(
                    (rng.randn(*p[k].shape) * 0.05).astype(np.float32))
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main,
                         args=(r, 2, port, q, init_pol, init_vf))
             for r in range(2)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(2):
        rank, status, out = q.get(timeout=240)
        assert status == 'ok', status
        results[rank] = out
    for p in procs:
        p.join(30)

    shards = [_make_shard(r) for r in range(2)]
    cat = {k: np.concatenate([s[k] for s in shards]) for k in shards[0]}
    batch = ob.OracleEpisodeBatch(max_episode_length=P, **cat)
    oracle = OraclePPO(init_pol, init_vf, max_episode_length=P,
                       max_optimization_epochs=3, minibatch_size=None,
                       policy_lr=1e-3, vf_lr=1e-3)
    want = oracle.train_once(batch)
    wp, wv = oracle.state()
    for rank in (0, 1):
        got = results[rank]
        for k, v in wp.items():
            assert np.allclose(got[k], v, atol=2e-6), (rank, k)
        for k, v in wv.items():
            assert np.allclose(got['vf:' + k], v, atol=2e-6), (rank, k)
        for k in ('policy/LossBefore', 'policy/LossAfter', 'policy/KL',
                  'vf/LossBefore', 'vf/LossAfter'):
            assert np.isclose(got['tab'][k], want[k], atol=2e-5, rtol=2e-5), \
                (rank, k, got['tab'][k], want[k])


@pytest.mark.timeout(300)
def test_two_rank_minibatch_ppo_with_unequal_shards_equals_oracle():
    """Ranks with different sample counts and the reference-default minibatch
    of 64 (``ppo.py:65-76``): both ranks take the same number of optimizer steps
    (``data_parallel_plan``: an even split into K minibatches per pass, one
    all-reduce each), every step's gradient is the mean over the union of the
    ranks' k-th minibatches, and the result equals the single-process oracle
    fed those unions as explicit id arrays."""
    from garage_amd.optimizers import data_parallel_plan
    from oracle import batch as ob
    from oracle import networks as nets
    from oracle.ppo import OraclePPO
    rng = np.random.RandomState(0)
    init_pol = nets.init_gaussian_mlp(rng, nets.POLICY_PREFIX, O, A, (16, 16),
                                      min_std=1e-6)
    init_vf = nets.init_gaussian_mlp(rng, nets.VALUE_PREFIX, O, 1, (16, 16))
    n_eps, mb, E = (41, 47), 64, 3
    shards = [_make_shard(r, n_eps[r]) for r in range(2)]
    counts = [int(s['lengths'].sum()) for s in shards]
    K = data_parallel_plan(counts, mb, 0)[0]
    # the case the old ceil(n / ceil(n / K)) rule got wrong on one rank
    assert counts[0] != counts[1] and K >= 4
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main,
                         args=(r, 2, port, q, init_pol, init_vf, 'ppo', 'gloo',
                               mb, n_eps))
             for r in range(2)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(2):
        rank, status, out = q.get(timeout=240)
        assert status == 'ok', status
        results[rank] = out
    for p in procs:
        p.join(30)
    for rank in (0, 1):
        assert results[rank]['adam_steps'] == (E * K, E * K)

    # the union of the ranks' k-th minibatches as ids of the concatenated batch
    ids = {'policy': [[] for _ in range(E * K)], 'vf': [[] for _ in range(E * K)]}
    base = 0
    for r, n in enumerate(counts):
        np.random.seed(50 + r)
        for which in ('policy', 'vf'):  # policy draws first
            perm = np.arange(n)
            np.random.shuffle(perm)
            for e in range(E):
                for k in range(K):
                    ids[which][e * K + k].append(
                        base + perm[k * n // K:(k + 1) * n // K])
                np.random.shuffle(perm)
        base += n
    ids = {w: [np.concatenate(parts) for parts in v] for w, v in ids.items()}
    cat = {k: np.concatenate([s[k] for s in shards]) for k in shards[0]}
    batch = ob.OracleEpisodeBatch(max_episode_length=P, **cat)
    oracle = OraclePPO(init_pol, init_vf, max_episode_length=P,
                       max_optimization_epochs=E, minibatch_size=mb,
                       policy_lr=1e-3, vf_lr=1e-3)
    want = oracle.train_once(batch, minibatch_ids=ids)
    wp, wv = oracle.state()
    for rank in (0, 1):
        got = results[rank]
        for k, v in wp.items():
            assert np.allclose(got[k], v, atol=3e-6), (rank, k)
        for k, v in wv.items():
            assert np.allclose(got['vf:' + k], v, atol=3e-6), (rank, k)
        for k in ('policy/LossBefore', 'policy/LossAfter', 'policy/KL',
                  'vf/LossBefore', 'vf/LossAfter'):
            assert np.isclose(got['tab'][k], want[k], atol=2e-5, rtol=2e-5), \
                (rank, k, got['tab'][k], want[k])
    for k in wp:  # replicas stay bit-identical
        assert np.array_equal(results[0][k], results[1][k]), k


class _TwinComm:
    """Stands in for a second rank that holds an identical shard: sums double,
    minima stay, counts are listed twice."""
    world_size = 2
    rank = 0
    group = None

    def all_reduce(self, tensor, op='sum'):
        if op == 'sum':
            tensor.mul_(2)
        return tensor

    def all_gather_int(self, value, device=None):
        return [int(value), int(value)]

    def broadcast(self, tensor, src=0):
        return tensor

    def barrier(self):
        pass


def _twin_algo(native, seed=0, hidden=(16, 16), minibatch=16):
    """PPO whose gradient exchange goes through the C++ epoch loop's all-reduce
    hook (``native``) or through the Python minibatch loop's ``grad_hook``."""
    import ctypes as C

    from garage_amd import _lib
    from garage_amd._dtypes import Box, EnvSpec
    from garage_amd.algos import PPO
    from garage_amd.optimizers import OptimizerWrapper
    from garage_amd.policies import GaussianMLPPolicy, GaussianMLPValueFunction
    spec = EnvSpec(Box(-np.inf, np.inf, (O, )), Box(-np.inf, np.inf, (A, )),
                   max_episode_length=P)
    torch.manual_seed(seed)
    pol = GaussianMLPPolicy(spec, hidden_sizes=hidden)
    vf = GaussianMLPValueFunction(spec, hidden_sizes=hidden)
    opt = (torch.optim.Adam, dict(lr=1e-3))
    algo = PPO(env_spec=spec, policy=pol, value_function=vf, sampler=None,
               policy_optimizer=OptimizerWrapper(opt, pol, 2, minibatch,
                                                 permutation='device', seed=5),
               vf_optimizer=OptimizerWrapper(opt, vf, 2, minibatch,
                                             permutation='device', seed=6))
    comm = _TwinComm()
    algo._comm = comm
    keep = []
    for o in (algo._policy_optimizer, algo._vf_optimizer):
        o.grad_hook = comm.all_reduce
        if native:

            class Handle:
                handle = C.c_void_p(1)  # opaque to the hook below
                world_size = 2

            o.native_comm = Handle()
    if native:
        lib = _lib.load()

        @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
        def twin_sum(_comm, buf, n, stream):
            # buf <- buf + buf: the sum over two ranks with identical gradients
            return lib.ga_axpby_f32(1.0, C.c_void_p(buf), 1.0, C.c_void_p(buf),
                                    n, C.c_void_p(stream))

        lib.ga_set_allreduce_hook(twin_sum)
        keep.append(twin_sum)
    return spec, pol, vf, algo, keep


@pytest.mark.parametrize('overlap', [True, False])
def test_native_dp_branch_equals_python_loop_with_a_twin_rank(overlap):
    """The data-parallel branch of ``ga_update_epoch*`` (scaled slab reduction
    -> all-reduce hook -> Adam, on one or two streams) against the Python
    minibatch loop with the same exchange: same kernels, same bits."""
    from garage_amd._dtypes import EpisodeBatch, StepType
    d = _make_shard(0)
    outs = []
    for native in (True, False):
        spec, pol, vf, algo, keep = _twin_algo(native)
        algo.overlap_updates = overlap
        st = np.asarray([StepType(int(s)) for s in d['step_types']],
                        dtype=object)
        batch = EpisodeBatch(env_spec=spec, episode_infos={},
                             observations=d['observations'],
                             last_observations=d['last_observations'],
                             actions=d['actions'], rewards=d['rewards'],
                             env_infos={}, agent_infos={}, step_types=st,
                             lengths=d['lengths'])
        assert algo._native_update_ok() == native
        algo._train_once(0, batch)
        outs.append((pol.net.params.clone(), vf.net.params.clone(),
                     dict(algo.last_tabular)))
        del keep
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])
    assert outs[0][2] == outs[1][2]


@pytest.mark.parametrize('overlap', [True, False])
@pytest.mark.parametrize('hidden', [(64, 64), (128, 128), (256, 256), (96, 256)])
def test_native_dp_branch_of_the_fused_step_kernels_with_a_twin_rank(hidden, overlap):
    """What the multi-GPU bench runs: the FUSED optimizer-step kernels (one-launch
    narrow step at 2 x 64; first layer + last hidden layer + head + loss, data
    gradient + first-layer weight gradient, region reduction WITHOUT Adam at
    128 / 256; (96, 256): the first layer as its own launch) -> all-reduce hook ->
    Adam, inside the C++ epoch loops on one or two streams, against the Python
    minibatch loop with the same exchange (``phase = 1`` + ``grad_hook`` +
    ``adam_step``): same kernels, same bits.  The fused kernels are really taken
    (switching them off changes the bits)."""
    from garage_amd import _lib
    from garage_amd._dtypes import EpisodeBatch, StepType
    lib = _lib.load()
    d = _make_shard(0, 60)
    st = np.asarray([StepType(int(s)) for s in d['step_types']], dtype=object)

    def run(native, fused=True):
        spec, pol, vf, algo, keep = _twin_algo(native, hidden=hidden,
                                               minibatch=100)
        algo.overlap_updates = overlap
        batch = EpisodeBatch(env_spec=spec, episode_infos={},
                             observations=d['observations'],
                             last_observations=d['last_observations'],
                             actions=d['actions'], rewards=d['rewards'],
                             env_infos={}, agent_infos={}, step_types=st,
                             lengths=d['lengths'])
        assert algo._native_update_ok() == native
        try:
            lib.ga_set_fused_train(1 if fused else 0)
            algo._train_once(0, batch)
        finally:
            lib.ga_set_fused_train(1)
        del keep
        return (pol.net.params.clone(), vf.net.params.clone(),
                dict(algo.last_tabular), (pol.net.adam_steps, vf.net.adam_steps))

    native, python, unfused = run(True), run(False), run(True, fused=False)
    assert torch.equal(native[0], python[0]) and torch.equal(native[1], python[1])
    assert native[2] == python[2] and native[3] == python[3]
    assert native[3][0] > 2  # several minibatches per pass
    assert not torch.equal(native[0], unfused[0])
    assert float((native[0] - unfused[0]).abs().max()) < 5e-4


@pytest.mark.timeout(300)
def test_two_rank_trpo_equals_single_process_oracle():
    """The conjugate-gradient policy step with the batch sharded over two ranks
    (share-weighted gradient / Fisher-product / loss / KL sums, replicated CG
    vectors) against the oracle's TRPO on the concatenated batch."""
    from oracle import batch as ob
    from oracle import networks as nets
    from oracle.trpo import OracleTRPO
    rng = np.random.RandomState(0)
    init_pol = nets.init_gaussian_mlp(rng, nets.POLICY_PREFIX, O, A, (16, 16),
                                      min_std=1e-6)
    init_vf = nets.init_gaussian_mlp(rng, nets.VALUE_PREFIX, O, 1, (16, 16))
    for p in (init_pol, init_vf):
        for k in p:
            if 'min_std' not in k:
                p[k] = p[k] + torch.from_numpy(
                    (rng.randn(*p[k].shape) * 0.05).astype(np.float32))
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main,
                         args=(r, 2, port, q, init_pol, init_vf, 'trpo'))
             for r in range(2)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(2):
        rank, status, out = q.get(timeout=240)
        assert status == 'ok', status
        results[rank] = out
    for p in procs:
        p.join(30)
    shards = [_make_shard(r) for r in range(2)]
    cat = {k: np.concatenate([s[k] for s in shards]) for k in shards[0]}
    batch = ob.OracleEpisodeBatch(max_episode_length=P, **cat)
    oracle = OracleTRPO(init_pol, init_vf, max_episode_length=P,
                        max_optimization_epochs=3, minibatch_size=None,
                        vf_lr=1e-3)
    want = oracle.train_once(batch)
    wp, wv = oracle.state()
    dscale = np.abs(oracle.cg.trace['descent_step']).max()
    for rank in (0, 1):
        got = results[rank]
        assert got['accepted'] == oracle.cg.trace['accepted']
        for k, v in wp.items():
            assert np.allclose(got[k], v, atol=2e-3 * dscale), (rank, k)
        for k, v in wv.items():
            assert np.allclose(got['vf:' + k], v, atol=2e-6), (rank, k)
        for k in ('policy/LossBefore', 'policy/LossAfter', 'policy/KL',
                  'vf/LossBefore', 'vf/LossAfter'):
            assert np.isclose(got['tab'][k], want[k], atol=2e-5, rtol=1e-3), \
                (rank, k, got['tab'][k], want[k])
    # both ranks hold the same parameters bit for bit
    for k in wp:
        assert np.array_equal(results[0][k], results[1][k]), k


@pytest.mark.timeout(300)
def test_rccl_communicators_of_one_rank_on_two_streams():
    """``ga_comm_*`` against the real RCCL library with a world of one rank (all a
    one-GPU box allows): unique id -> init -> in-place sum on a side stream and
    on the current stream with two communicators, as the overlapped policy /
    value passes use them -> destroy.  A one-rank sum returns its input."""
    import ctypes as C

    from garage_amd import _lib
    lib = _lib.load()
    dev = torch.device('cuda')
    comms = []
    for _ in range(2):
        raw = (C.c_ubyte * 128)()
        assert lib.ga_comm_unique_id(raw) == 0, lib.ga_last_error().decode()
        h = lib.ga_comm_init_rank(raw, 0, 1)
        assert h, lib.ga_last_error().decode()
        comms.append(h)
    side = torch.cuda.Stream()
    a = torch.randn(71943, device=dev)
    b = torch.randn(70658, device=dev)
    a0, b0 = a.clone(), b.clone()
    torch.cuda.synchronize()
    for _ in range(3):
        assert lib.ga_comm_allreduce_sum_f32(
            comms[0], _lib.dptr(a), a.numel(), _lib.stream_ptr()) == 0
        with torch.cuda.stream(side):
            assert lib.ga_comm_allreduce_sum_f32(
                comms[1], _lib.dptr(b), b.numel(), _lib.stream_ptr()) == 0
    torch.cuda.synchronize()
    assert torch.equal(a, a0) and torch.equal(b, b0)
    for h in comms:
        assert lib.ga_comm_destroy(h) == 0


@pytest.mark.timeout(300)
def test_one_rank_nccl_group_takes_the_rccl_branch_of_the_epoch_loop():
    """``shard_algo`` over a real ``nccl`` (= RCCL) process group of one rank: the
    library-owned communicators are created from the broadcast unique id, the
    C++ epoch loop takes its data-parallel branch (scaled slab sum -> RCCL
    all-reduce -> Adam) and the iteration equals the single-process oracle."""
    from oracle import batch as ob
    from oracle import networks as nets
    from oracle.ppo import OraclePPO
    rng = np.random.RandomState(0)
    init_pol = nets.init_gaussian_mlp(rng, nets.POLICY_PREFIX, O, A, (16, 16),
                                      min_std=1e-6)
    init_vf = nets.init_gaussian_mlp(rng, nets.VALUE_PREFIX, O, 1, (16, 16))
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    proc = ctx.Process(target=_rank_main,
                       args=(0, 1, port, q, init_pol, init_vf, 'ppo', 'nccl'))
    proc.start()
    rank, status, got = q.get(timeout=240)
    proc.join(30)
    assert status == 'ok', status
    assert got['native_comm'] and got['native_loop']
    batch = ob.OracleEpisodeBatch(max_episode_length=P, **_make_shard(0))
    oracle = OraclePPO(init_pol, init_vf, max_episode_length=P,
                       max_optimization_epochs=3, minibatch_size=None,
                       policy_lr=1e-3, vf_lr=1e-3)
    want = oracle.train_once(batch)
    wp, wv = oracle.state()
    for k, v in wp.items():
        assert np.allclose(got[k], v, atol=2e-6), k
    for k, v in wv.items():
        assert np.allclose(got['vf:' + k], v, atol=2e-6), k
    for k in ('policy/LossBefore', 'policy/LossAfter', 'policy/KL',
              'vf/LossBefore', 'vf/LossAfter'):
        assert np.isclose(got['tab'][k], want[k], atol=2e-5, rtol=2e-5), k


@pytest.mark.timeout(300)
@pytest.mark.parametrize('hidden', [(64, 64), (256, 256)])
def test_one_rank_rccl_through_the_fused_step_kernels(hidden):
    """The bench's multi-GPU configuration as far as one GPU allows: a real ``nccl``
    (= RCCL) process group of one rank, the library-owned communicators, the fused
    optimizer-step kernels with minibatches on two streams, one real
    ``ncclAllReduce`` per optimizer step and network.  A one-rank sum is the
    identity, so the iteration must equal the same iteration without any process
    group."""
    from garage_amd._dtypes import Box, EnvSpec, EpisodeBatch, StepType
    from garage_amd.algos import PPO
    from garage_amd.optimizers import OptimizerWrapper
    from garage_amd.policies import GaussianMLPPolicy, GaussianMLPValueFunction
    from oracle import networks as nets
    rng = np.random.RandomState(1)
    init_pol = nets.init_gaussian_mlp(rng, nets.POLICY_PREFIX, O, A, hidden,
                                      min_std=1e-6)
    init_vf = nets.init_gaussian_mlp(rng, nets.VALUE_PREFIX, O, 1, hidden)
    n_eps = (60, )
    # data parallel runs split a pass EVENLY into K minibatches (every rank must
    # take the same number of steps); a minibatch size that divides the sample
    # count makes that the same split as the single-process one
    S = int(_make_shard(0, n_eps[0])['lengths'].sum())
    mb = next(S // k for k in (4, 3, 5, 6, 2, 1) if S % k == 0)
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    proc = ctx.Process(target=_rank_main,
                       args=(0, 1, port, q, init_pol, init_vf, 'ppo', 'nccl', mb,
                             n_eps, hidden))
    proc.start()
    rank, status, got = q.get(timeout=240)
    proc.join(30)
    assert status == 'ok', status
    assert got['native_comm'] and got['native_loop']
    # the same iteration in this process, no process group
    spec = EnvSpec(Box(-np.inf, np.inf, (O, )), Box(-np.inf, np.inf, (A, )),
                   max_episode_length=P)
    pol = GaussianMLPPolicy(spec, hidden_sizes=hidden)
    vf = GaussianMLPValueFunction(spec, hidden_sizes=hidden)
    pol.load_state_dict(init_pol)
    vf.load_state_dict(init_vf)
    opt = (torch.optim.Adam, dict(lr=1e-3))
    algo = PPO(env_spec=spec, policy=pol, value_function=vf, sampler=None,
               policy_optimizer=OptimizerWrapper(opt, pol, 3, mb),
               vf_optimizer=OptimizerWrapper(opt, vf, 3, mb))
    d = _make_shard(0, n_eps[0])
    np.random.seed(50)
    batch = EpisodeBatch(env_spec=spec, episode_infos={},
                         observations=d['observations'],
                         last_observations=d['last_observations'],
                         actions=d['actions'], rewards=d['rewards'], env_infos={},
                         agent_infos={},
                         step_types=np.asarray([StepType(int(s))
                                                for s in d['step_types']],
                                               dtype=object),
                         lengths=d['lengths'])
    algo._train_once(0, batch)
    assert got['adam_steps'] == (pol.net.adam_steps, vf.net.adam_steps)
    assert got['adam_steps'][0] == 3 * (S // mb)
    for k, v in pol.state_dict().items():
        assert np.allclose(got[k], v.numpy(), atol=1e-6), k
    for k, v in vf.state_dict().items():
        assert np.allclose(got['vf:' + k], v.numpy(), atol=1e-6), k
    for k, v in algo.last_tabular.items():
        assert np.isclose(got['tab'][k], v, atol=1e-6, rtol=1e-6), k
