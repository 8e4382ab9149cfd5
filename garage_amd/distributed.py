"""Data parallelism over the GPUs of one node: env shards + RCCL all-reduce.

The reference has no collective on this path (LocalSampler
is single process).  The new DP dimension follows section 8e: rank ``r`` owns
environments ``[r*n, (r+1)*n)`` and full replicas of policy, value function
and Adam state; rollout, baselines and the GAE scan are rank local; the only
exchanges are
  1. the advantage moments (sum / count, squared deviations, min) so that
     ``center_adv`` / ``positive_adv`` see the global batch,
  2. one all-reduce(mean) of the flat gradient buffer per optimizer step
     (policy 0.29 MB, value 0.28 MB at C3: latency bound, so one buffer, one
     call), and
  3. the logged scalars.
``torch.distributed`` backend ``nccl`` is RCCL on ROCm (xGMI inside a node);
``gloo`` runs the same logic on CPU tensors for the world_size-2 tests.
"""
import os

import torch
import torch.distributed as dist


class Comm:
    """Thin wrapper so algorithms do not depend on torch.distributed directly."""

    def __init__(self, group=None):
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def all_reduce(self, tensor, op='sum'):
        ops = {'sum': dist.ReduceOp.SUM, 'min': dist.ReduceOp.MIN,
               'max': dist.ReduceOp.MAX}
        dist.all_reduce(tensor, op=ops[op], group=self.group)
        return tensor

    def all_reduce_mean(self, tensor):
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.group)
        tensor.mul_(1.0 / self.world_size)
        return tensor

    def all_gather_int(self, value, device=None):
        """One integer per rank, as a Python list (sample counts etc.)."""
        t = torch.zeros(self.world_size, dtype=torch.int64,
                        device=device or ('cuda' if dist.get_backend(
                            self.group) == 'nccl' else 'cpu'))
        t[self.rank] = int(value)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return [int(v) for v in t.cpu().tolist()]

    def broadcast(self, tensor, src=0):
        dist.broadcast(tensor, src=src, group=self.group)
        return tensor

    def barrier(self):
        dist.barrier(group=self.group)


def init_from_env(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_*."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1:
        return None
    local_rank = int(os.environ.get('LOCAL_RANK', os.environ.get('RANK', '0')))
    if backend is None:
        backend = os.environ.get(
            'GARAGE_AMD_BACKEND',
            'nccl' if torch.cuda.is_available() else 'gloo')
    if torch.cuda.is_available():
        # one rank per GPU; with fewer GPUs than ranks (a gloo rehearsal on a
        # one-GPU box) ranks share devices round-robin
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
    if not dist.is_initialized():
        dist.init_process_group(backend=backend)
    return Comm()


class NativeComm:
    """RCCL communicator owned by ``libgarage_amd`` (``ga_comm_*``), so the C++
    epoch loop can all-reduce gradients without returning to Python.  The
    128-byte unique id is created on rank 0 and broadcast with torch.

    Construction is collective and so is its failure: after each step that can
    fail locally (creating the id on rank 0, ``ncclCommInitRank`` on every rank)
    the ranks all-reduce an ok flag, so either every rank holds a communicator
    or every rank raises -- no rank is left waiting in a broadcast its peer
    never enters.
    """

    def __init__(self, comm):
        import ctypes as C

        from garage_amd import _lib
        lib = _lib.load()
        raw = (C.c_ubyte * 128)()
        # ncclCommInitRank is itself collective: a rank that cannot even load
        # librccl must be known before anybody enters it
        self._agree(comm, bool(lib.ga_comm_available()), 'loading librccl')
        err = ''
        if comm.rank == 0 and lib.ga_comm_unique_id(raw) != 0:
            err = lib.ga_last_error().decode()
        self._agree(comm, not err, 'ga_comm_unique_id: ' + err)
        t = torch.tensor(list(raw), dtype=torch.uint8, device='cuda')
        comm.broadcast(t, src=0)
        raw = (C.c_ubyte * 128)(*t.cpu().tolist())
        self.handle = lib.ga_comm_init_rank(raw, comm.rank, comm.world_size)
        err = '' if self.handle else lib.ga_last_error().decode()
        try:
            self._agree(comm, bool(self.handle), 'ga_comm_init_rank: ' + err)
        except RuntimeError:
            if self.handle:
                lib.ga_comm_destroy(self.handle)
                self.handle = None
            raise
        self.world_size = comm.world_size
        self.rank = comm.rank
        self.rccl_ranks = int(lib.ga_comm_count(self.handle))

    def destroy(self):
        """Release the communicator (collective in RCCL: call on every rank)."""
        if getattr(self, 'handle', None):
            from garage_amd import _lib
            _lib.load().ga_comm_destroy(self.handle)
            self.handle = None

    @staticmethod
    def _agree(comm, ok, what):
        flag = torch.tensor([1.0 if ok else 0.0], device='cuda')
        comm.all_reduce(flag, 'min')
        if float(flag.item()) < 1.0:
            raise RuntimeError('native RCCL communicator: a rank failed in ' +
                               what)


def _make_native_comm(comm):
    """Seam for tests (a world of one GPU cannot make RCCL fail half way)."""
    return NativeComm(comm)


def combine_moments(stats, comm):
    """Global (sum, count) / squared deviations / min from per-rank stats.

    ``stats`` is the 4-slot fp64 tensor of ``ga_stats_f32``; used between the
    stages of :func:`garage_amd.engine.center_advantages`.
    """
    comm.all_reduce(stats[0:2], 'sum')
    return stats


def shard_algo(algo, comm):
    """Make ``algo`` (VPG / PPO) data parallel over ``comm``.

    Parameters and Adam state are broadcast from rank 0; every optimizer step
    then averages the flat gradient buffer across ranks, and the advantage
    normalisation / logged scalars use global statistics.
    """
    if comm is None:
        return algo
    algo._comm = comm
    use_rccl = dist.get_backend(comm.group) == 'nccl'
    pairs = ((algo.policy, algo._policy_optimizer),
             (algo._value_function, algo._vf_optimizer))
    for module, opt in pairs:
        comm.broadcast(module.net.params)
        comm.broadcast(module.net.exp_avg)
        comm.broadcast(module.net.exp_avg_sq)
        # the local gradient is pre-scaled by S_local / S_global
        # (OptimizerWrapper.dp_grad_scale), so the exchange is a plain sum
        opt.grad_hook = comm.all_reduce
        opt.native_comm = None
    # RCCL inside the C++ epoch loop; one communicator per network because the
    # two passes run on two streams and a communicator's collectives must be
    # issued in one order on every rank.  All or none: either BOTH networks of
    # EVERY rank hold a communicator, or both take the Python minibatch loop
    # with torch.distributed all-reduces (NativeComm raises on every rank or
    # on none, so the decision is the same everywhere) -- a mixed state would
    # run one network's steps through the C++ loop and the other's through
    # Python, and report neither correctly.
    if use_rccl and os.environ.get('GARAGE_AMD_NATIVE_COMM', '1') != '0':
        made = []
        try:
            for _ in pairs:
                made.append(_make_native_comm(comm))
        except RuntimeError as exc:
            # fall back to the Python-driven minibatch loop, which all-reduces
            # through torch.distributed (same results, more host overhead)
            import warnings
            for c in made:
                c.destroy()
            made = []
            warnings.warn('native RCCL communicator unavailable ({}); using '
                          'torch.distributed for the gradient all-reduce of '
                          'both networks'.format(exc))
        for (_, opt), c in zip(pairs, made):
            opt.native_comm = c
    algo._old_policy.sync(algo.policy)
    return algo


def gradient_exchange(algo):
    """How ``algo``'s optimizer steps exchange gradients, for run records:
    ``(description, rccl_ranks)``; ``rccl_ranks`` is what RCCL itself reports
    for the library-owned communicators (``None`` without them)."""
    if getattr(algo, '_comm', None) is None:
        return 'none (one process)', None
    opts = (algo._policy_optimizer, algo._vf_optimizer)
    native = [getattr(o, 'native_comm', None) for o in opts]
    if all(n is not None for n in native):
        return ('rccl all-reduce inside the C++ epoch loop (one library-owned '
                'communicator per network)', min(n.rccl_ranks for n in native))
    backend = dist.get_backend(algo._comm.group)
    if any(n is not None for n in native):  # shard_algo never leaves this
        return ('MIXED: native communicator for {} only, torch.distributed '
                '({}) for the other network'.format(
                    'the policy' if native[0] is not None else
                    'the value function', backend), None)
    return ('torch.distributed all-reduce ({}) from the Python minibatch loop'
            ' (FALLBACK: no native communicator)'.format(backend), None)
