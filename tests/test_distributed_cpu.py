"""world_size-2 ``gloo`` tests of the data-parallel logic (CPU, no GPU needed).

The ranks use the product's ``garage_amd.distributed.Comm`` exactly as the GPU
path does (advantage-moment exchange between the reduction stages, all-reduce
mean of the flat gradient buffer per optimizer step, scalar averaging); the
per-rank arithmetic comes from the oracle, and the result must equal the
single-process oracle on the concatenated batch.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _run(fn, world=2):
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_entry, args=(fn, r, world, port, q))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(30)
    for r in results:
        assert r[1] == 'ok', r
    return results


def _entry(fn, rank, world, port, q):
    try:
        os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank),
                          WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                          MASTER_PORT=str(port))
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        from garage_amd.distributed import init_from_env
        comm = init_from_env(backend='gloo')
        assert comm.world_size == world and comm.rank == rank
        fn(comm)
        import torch.distributed as dist
        dist.destroy_process_group()
        q.put((rank, 'ok'))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, 'error: %r\n%s' % (e, traceback.format_exc())))


def _moments(comm):
    """Staged global centring == torch on the concatenated shards (vpg.py:371-377)."""
    rng = np.random.RandomState(0)
    full = torch.from_numpy((rng.randn(1001) * 2 + 0.5).astype(np.float32))
    shard = full[:600] if comm.rank == 0 else full[600:]
    stats = torch.zeros(4, dtype=torch.float64)
    stats[0], stats[1] = shard.double().sum(), shard.numel()
    comm.all_reduce(stats[0:2], 'sum')           # stage 1: sum, count
    mean = stats[0] / stats[1]
    stats[2] = ((shard.double() - mean)**2).sum()
    comm.all_reduce(stats[2:3], 'sum')           # stage 2: squared deviations
    var = stats[2] / (stats[1] - 1)
    got = (shard - mean.float()) / (var.float() + 1e-8)
    want = (full - full.mean()) / (full.var() + 1e-8)
    want = want[:600] if comm.rank == 0 else want[600:]
    assert torch.allclose(got, want, atol=1e-5)
    stats[3] = got.double().min()
    comm.all_reduce(stats[3:4], 'min')           # positive_adv
    assert np.isclose(stats[3].item(), ((full - full.mean()) /
                                        (full.var() + 1e-8)).min().item(),
                      atol=1e-5)


def _grad_average(comm):
    """all-reduce(mean) of per-rank minibatch gradients == gradient of the loss
    on the union (equal shard sizes), for the PPO surrogate and the value NLL."""
    from collections import OrderedDict

    from oracle import networks as nets
    rng = np.random.RandomState(1)
    pol = nets.init_gaussian_mlp(rng, nets.POLICY_PREFIX, 5, 3, (16, 16),
                                 min_std=1e-6)
    M = 64
    obs = torch.from_numpy(rng.randn(2 * M, 5).astype(np.float32))
    act = torch.from_numpy(rng.randn(2 * M, 3).astype(np.float32))
    adv = torch.from_numpy(rng.randn(2 * M).astype(np.float32))

    def loss_grad(o, a, ad):
        p = OrderedDict((k, v.clone().requires_grad_('min_std' not in k))
                        for k, v in pol.items())
        dist = nets.gaussian_dist(p, nets.POLICY_PREFIX, o)
        with torch.no_grad():
            old = nets.gaussian_dist(pol, nets.POLICY_PREFIX, o).log_prob(a) + 0.1
        ratio = (dist.log_prob(a) - old).exp()
        obj = torch.min(ratio * ad, torch.clamp(ratio, 0.8, 1.2) * ad)
        (-obj.mean()).backward()
        return torch.cat([p[k].grad.reshape(-1) for k in nets.trainable_keys(p)])

    sl = slice(0, M) if comm.rank == 0 else slice(M, 2 * M)
    flat = loss_grad(obs[sl], act[sl], adv[sl])
    comm.all_reduce_mean(flat)
    want = loss_grad(obs, act, adv)
    assert torch.allclose(flat, want, atol=1e-6)


def _broadcast(comm):
    t = torch.full((7, ), float(comm.rank + 1))
    comm.broadcast(t, src=0)
    assert torch.equal(t, torch.ones(7))
    comm.barrier()


@pytest.mark.timeout(180)
def _all_or_none(comm):
    """shard_algo's native-communicator decision, two gloo ranks: the second
    communicator fails (on every rank, as NativeComm's own handshake makes it)
    -> the first one is released and BOTH networks fall back."""
    import types
    import warnings

    import torch.distributed as dist

    from garage_amd import distributed as D

    class Net:

        def __init__(self):
            self.params = torch.full((8, ), float(comm.rank))
            self.exp_avg = torch.zeros(8)
            self.exp_avg_sq = torch.zeros(8)

    class Old:
        synced = False

        def sync(self, policy):
            self.synced = True

    def make_algo():
        mod = lambda: types.SimpleNamespace(net=Net())  # noqa: E731
        opt = lambda: types.SimpleNamespace(grad_hook=None)  # noqa: E731
        return types.SimpleNamespace(
            policy=mod(), _value_function=mod(), _policy_optimizer=opt(),
            _vf_optimizer=opt(), _old_policy=Old(), _comm=None)

    made = []

    class Fake:

        def __init__(self):
            self.alive, self.rccl_ranks = True, comm.world_size
            made.append(self)

        def destroy(self):
            self.alive = False

    real_backend, real_make = dist.get_backend, D._make_native_comm
    dist.get_backend = lambda group=None: 'nccl'
    try:
        # (1) the second construction fails
        calls = []

        def flaky(c):
            calls.append(1)
            if len(calls) == 2:
                raise RuntimeError('native RCCL communicator: a rank failed '
                                   'in ga_comm_init_rank: test')
            return Fake()

        D._make_native_comm = flaky
        algo = make_algo()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            D.shard_algo(algo, comm)
        assert len(calls) == 2 and len(made) == 1 and not made[0].alive
        assert algo._policy_optimizer.native_comm is None
        assert algo._vf_optimizer.native_comm is None
        assert any('both networks' in str(x.message) for x in w)
        assert algo._policy_optimizer.grad_hook is not None
        text, ranks = D.gradient_exchange(algo)
        assert 'FALLBACK' in text and ranks is None
        # rank 0's parameters were broadcast either way
        assert float(algo.policy.net.params.sum()) == 0.0
        assert algo._old_policy.synced
        # (2) both succeed -> both native
        D._make_native_comm = lambda c: Fake()
        algo = make_algo()
        D.shard_algo(algo, comm)
        assert algo._policy_optimizer.native_comm.alive
        assert algo._vf_optimizer.native_comm.alive
        assert algo._policy_optimizer.native_comm is not \
            algo._vf_optimizer.native_comm
        text, ranks = D.gradient_exchange(algo)
        assert text.startswith('rccl all-reduce') and ranks == comm.world_size
        # (3) a hand-made mixed state is reported as such, never as a fallback
        algo._vf_optimizer.native_comm = None
        text, ranks = D.gradient_exchange(algo)
        assert text.startswith('MIXED') and ranks is None
    finally:
        dist.get_backend, D._make_native_comm = real_backend, real_make


def test_native_communicators_are_all_or_none_two_ranks():
    _run(_all_or_none)


def test_global_advantage_moments_two_ranks():
    _run(_moments)


@pytest.mark.timeout(180)
def test_gradient_average_two_ranks():
    _run(_grad_average)


@pytest.mark.timeout(180)
def test_broadcast_and_barrier_two_ranks():
    _run(_broadcast)


@pytest.mark.timeout(240)
def test_bench_starts_its_own_ranks_and_fails_loudly_without_a_gpu():
    """``python bench.py --gpus 2`` with no launcher around it starts two ranks
    itself (torchrun-style environment, gloo here) -- and since the product has
    no CPU fallback each rank dies in ``require_gpu``, which the parent must
    report as a non-zero exit, not as an ``n_gpus: 1`` line."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'),
                          '--gpus', '2', '--steps', '1', '--warmup', '0'],
                         cwd=root, env=env, capture_output=True, text=True,
                         timeout=200)
    assert out.returncode != 0
    assert 'rank 0 exited with code' in out.stderr or \
        'rank 1 exited with code' in out.stderr
    assert 'no CPU fallback' in out.stderr
    assert '"n_gpus"' not in out.stdout
