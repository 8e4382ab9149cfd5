"""``OptimizerWrapper`` with a fused device Adam and minibatch id streams.

Mirrors ``garage.torch.optimizers.OptimizerWrapper``
(``torch/optimizers/optimizer_wrapper.py:6-63``), ``make_optimizer``
(``_functions.py:25-65``) and ``BatchDataset``
(``np/optimizers/minibatch_dataset.py:4-35``).
"""
import numpy as np
import torch

from garage_amd._lib import call, dptr, stream_ptr


class ConjugateGradientOptimizer:
    """Hyper-parameters of garage's constrained optimizer
    (``torch/optimizers/conjugate_gradient_optimizer.py:107-145``).

    ``OptimizerWrapper((ConjugateGradientOptimizer,
    dict(max_constraint_value=0.01)), policy)`` selects the conjugate-gradient /
    backtracking policy step that :class:`garage_amd.algos.TRPO` runs on the
    device; this class only carries the settings (same names and defaults).
    """

    def __init__(self, params=None, max_constraint_value=None, cg_iters=10,
                 max_backtracks=15, backtrack_ratio=0.8, hvp_reg_coeff=1e-5,
                 accept_violation=False):
        del params
        if max_constraint_value is None:
            raise TypeError("__init__() missing 1 required positional "
                            "argument: 'max_constraint_value'")
        self.state = dict(max_constraint_value=max_constraint_value,
                          cg_iters=cg_iters, max_backtracks=max_backtracks,
                          backtrack_ratio=backtrack_ratio,
                          hvp_reg_coeff=hvp_reg_coeff,
                          accept_violation=accept_violation)


def _parse_optimizer(optimizer):
    """``make_optimizer``: a type, or ``(type, kwargs)``."""
    kwargs = {}
    if isinstance(optimizer, tuple):
        opt_type, kwargs = optimizer
        kwargs = dict(kwargs)
    else:
        opt_type = optimizer
    name = getattr(opt_type, '__name__', str(opt_type))
    if name == 'ConjugateGradientOptimizer':
        hyper = ConjugateGradientOptimizer(**kwargs).state
        hyper['kind'] = 'cg'
        return hyper
    return _torch_optimizer_hyper(name, kwargs)


def _torch_optimizer_hyper(name, kwargs):
    """Settings of a ``torch.optim`` class as the kernels take them.  ``kind``
    ``'adam'`` is torch's default-shaped Adam (no weight decay, no amsgrad): the
    one fused into the native update loops.  Everything else is ``'generic'`` --
    one elementwise launch per step (``ga_optimizer_step_f32``) behind the
    per-minibatch Python loop -- with ``code`` / ``h`` / ``flags`` as the header
    documents them.  Same keyword names and defaults as torch."""
    kw = dict(kwargs)

    def take(key, default):
        return kw.pop(key, default)

    if name in ('Adam', 'AdamW'):
        lr = take('lr', 1e-3)
        betas = tuple(take('betas', (0.9, 0.999)))
        eps = take('eps', 1e-8)
        wd = take('weight_decay', 1e-2 if name == 'AdamW' else 0)
        amsgrad = bool(take('amsgrad', False))
        for k in ('foreach', 'capturable', 'differentiable', 'fused'):
            kw.pop(k, None)
        if take('maximize', False) or kw:
            raise NotImplementedError(
                'torch.optim.{} option(s) {} are not supported'.format(
                    name, sorted(kw) or ['maximize']))
        if name == 'Adam' and not wd and not amsgrad:
            return dict(kind='adam', lr=lr, betas=betas, eps=eps)
        return dict(kind='generic', name=name, code=3, lr=lr, betas=betas,
                    eps=eps, h=[lr, betas[0], betas[1], eps, wd],
                    flags=int(amsgrad) | (2 if name == 'AdamW' else 0),
                    needs=(True, True, amsgrad))
    if name == 'SGD':
        lr = take('lr', 1e-3)
        momentum = take('momentum', 0)
        dampening = take('dampening', 0)
        wd = take('weight_decay', 0)
        nesterov = bool(take('nesterov', False))
        for k in ('foreach', 'differentiable', 'fused'):
            kw.pop(k, None)
        if take('maximize', False) or kw:
            raise NotImplementedError(
                'torch.optim.SGD option(s) {} are not supported'.format(
                    sorted(kw) or ['maximize']))
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError('Nesterov momentum requires a momentum and zero '
                             'dampening')  # torch/optim/sgd.py
        return dict(kind='generic', name=name, code=1, lr=lr,
                    h=[lr, momentum, dampening, wd, 0.0], flags=int(nesterov),
                    needs=(momentum != 0, False, False))
    if name == 'RMSprop':
        lr = take('lr', 1e-2)
        alpha = take('alpha', 0.99)
        eps = take('eps', 1e-8)
        wd = take('weight_decay', 0)
        momentum = take('momentum', 0)
        centered = bool(take('centered', False))
        for k in ('foreach', 'capturable', 'differentiable'):
            kw.pop(k, None)
        if take('maximize', False) or kw:
            raise NotImplementedError(
                'torch.optim.RMSprop option(s) {} are not supported'.format(
                    sorted(kw) or ['maximize']))
        return dict(kind='generic', name=name, code=2, lr=lr,
                    h=[lr, alpha, eps, wd, momentum], flags=int(centered),
                    needs=(True, momentum > 0, centered))
    raise NotImplementedError(
        'garage_amd implements torch.optim.Adam / AdamW / SGD / RMSprop (and '
        'garage\'s ConjugateGradientOptimizer); got {}'.format(name))


def data_parallel_plan(counts, minibatch_size, rank):
    """Minibatch plan of one pass when rank ``r`` holds ``counts[r]`` samples.

    Every rank must take the same number of optimizer steps (each issues one
    gradient all-reduce), whatever its own count: ``K = ceil(max(counts) /
    minibatch_size)`` minibatches per pass, rank ``r``'s minibatch ``k`` = ids
    ``[k * counts[r] // K, (k + 1) * counts[r] // K)`` of its permutation.
    Global minibatch ``k`` is the union of the ranks' ``k``-th minibatches
    (SURVEY.md section 8e), so its mean gradient weights rank ``r`` by
    ``rows_r,k / sum_r rows_r,k``.  Returns ``(K, scales[K] float32)`` for
    ``rank``; raises if some rank could not fill ``K`` minibatches.
    """
    counts = [int(c) for c in counts]
    K = -(-max(counts) // int(minibatch_size))
    if min(counts) < K:
        raise RuntimeError('a rank holds fewer samples than there are '
                           'minibatches per pass')
    sizes = np.asarray([[(k + 1) * c // K - k * c // K for k in range(K)]
                        for c in counts], dtype=np.float64)
    return K, (sizes[rank] / sizes.sum(axis=0)).astype(np.float32)


class OptimizerWrapper:
    """Adam over one module's flat parameter buffer + minibatch iteration.

    Args:
        optimizer: ``torch.optim.Adam`` or ``(torch.optim.Adam, {'lr': ...})``.
        module: a ``garage_amd`` policy / value function (has ``.net``).
        max_optimization_epochs (int): passes over the data per update.
        minibatch_size (int or None): ``None`` = one full batch, no shuffle.
        permutation (str): ``'numpy'`` draws the ids exactly like the reference
            (global ``np.random.shuffle`` on the host, cumulative, one extra
            shuffle per pass: SURVEY.md Q8) and ships them to the device;
            ``'device'`` evaluates a keyed Feistel permutation in a HIP kernel
            (no host work; a different but equally valid shuffle).
        seed (int): key of the ``'device'`` permutations.
    """

    def __init__(self, optimizer, module, max_optimization_epochs=1,
                 minibatch_size=None, permutation='numpy', seed=0):
        self._hyper = _parse_optimizer(optimizer)
        self._module = module
        self._max_optimization_epochs = max_optimization_epochs
        self._minibatch_size = minibatch_size
        if permutation not in ('numpy', 'device'):
            raise ValueError("permutation must be 'numpy' or 'device'")
        self._permutation = permutation
        self._seed = int(seed)
        self._draws = 0
        self.grad_hook = None  # multi-GPU: called on the flat grads
        self.native_comm = None  # RCCL handle for the C++ epoch loop
        # data parallel, set per iteration by the algorithm: minibatches per
        # pass (identical on every rank, or the collectives would not line up)
        # and this rank's share of the global sample count
        self.dp_minibatches = None
        self.dp_grad_scale = 1.0
        # per-minibatch shares (rows of this rank's minibatch k / rows of the
        # global minibatch k); None: dp_grad_scale for every step
        self.dp_grad_scales = None
        self._cur_grad_scale = None

    def __getstate__(self):
        state = self.__dict__.copy()
        state['grad_hook'] = None  # process-group objects do not pickle
        state['native_comm'] = None
        return state

    @property
    def net(self):
        return self._module.net

    def epoch_permutations(self, n):
        """Yield, per optimisation pass, the device int32 id order of that pass
        (``None`` when ``minibatch_size is None``: one full batch, no shuffle).

        ``'numpy'`` mode consumes the global numpy RNG exactly like
        ``BatchDataset``: one shuffle at construction, one after every pass,
        applied cumulatively to the same id array.
        """
        if self._minibatch_size is None:
            for _ in range(self._max_optimization_epochs):
                yield None
            return
        dev = self.net.device
        if self._permutation == 'numpy':
            ids = np.arange(n, dtype=np.int32)
            np.random.shuffle(ids)  # BatchDataset.__init__ -> update()
            for _ in range(self._max_optimization_epochs):
                yield torch.from_numpy(ids.copy()).to(dev)
                np.random.shuffle(ids)  # after each full pass
        else:
            for _ in range(self._max_optimization_epochs):
                perm = torch.empty(n, dtype=torch.int32, device=dev)
                self._draws += 1
                key = (self._seed * 0x9E3779B97F4A7C15 +
                       self._draws * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
                call('ga_permutation_i32', n, key, dptr(perm), stream_ptr())
                yield perm

    def minibatch_indices(self, n):
        """Yield one device int32 id tensor per minibatch (``None``: all rows).

        Same number, sizes and -- in ``'numpy'`` mode -- contents as the
        minibatches ``get_minibatch`` of the reference yields for ``n`` rows.
        """
        bounds = self.minibatch_bounds(n)
        for perm in self.epoch_permutations(n):
            self._cur_grad_scale = None
            if perm is None:
                yield None
                continue
            for k in range(len(bounds) - 1):
                if self.dp_grad_scales is not None:
                    self._cur_grad_scale = float(self.dp_grad_scales[k])
                yield perm[bounds[k]:bounds[k + 1]]

    def minibatch_bounds(self, n):
        """``b`` with minibatch ``k`` = ids ``[b[k], b[k+1])`` of a pass over
        ``n`` rows.  Single process: ``BatchDataset``'s ``ceil(n / mb)``
        minibatches of ``mb`` ids, the last one partial.  Data parallel: ranks
        hold different ``n`` but must take the same number of optimizer steps
        (one gradient all-reduce each), so the ids are split into exactly
        ``dp_minibatches`` parts at ``k * n // dp_minibatches``."""
        if self._minibatch_size is None:
            return [0, int(n)]
        if self.dp_minibatches:
            K = int(self.dp_minibatches)
            return [k * int(n) // K for k in range(K + 1)]
        mb = int(self._minibatch_size)
        return list(range(0, int(n), mb)) + [int(n)]

    def local_minibatch_size(self, n):
        """Rows of the largest minibatch on this rank (workspace sizing).
        Single process: ``minibatch_size``; data parallel:
        ``ceil(n / dp_minibatches)`` (see :meth:`minibatch_bounds`)."""
        if self._minibatch_size is None:
            return None
        if self.dp_minibatches:
            return -(-n // int(self.dp_minibatches))
        return int(self._minibatch_size)

    def get_minibatch(self, *inputs):
        """Reference-shaped generator: lists of gathered tensors."""
        n = inputs[0].shape[0]
        for idx in self.minibatch_indices(n):
            if idx is None:
                yield list(inputs)
            else:
                yield [d[idx.long()] for d in inputs]

    def zero_grad(self):
        self.net.grads.zero_()

    @property
    def default_adam(self):
        """torch's default-shaped Adam: the optimizer fused into the native update
        loops; anything else steps through ``FlatMLP.optimizer_step`` from the
        per-minibatch Python loop."""
        return self._hyper['kind'] == 'adam'

    def apply_step(self):
        """One optimizer step on ``net.grads`` (already reduced / exchanged)."""
        h = self._hyper
        if h['kind'] == 'adam':
            self.net.adam_step(h['lr'], h['betas'], h['eps'])
        else:
            self.net.optimizer_step(h)

    def step(self, **closure):
        """Reduce the gradient slabs, (all-reduce,) optimizer step."""
        del closure
        if self._hyper['kind'] == 'cg':
            raise NotImplementedError(
                'the conjugate-gradient step is driven by garage_amd.algos.TRPO')
        scale = 1.0
        if self.grad_hook is not None:
            scale = (self.dp_grad_scale if self._cur_grad_scale is None
                     else self._cur_grad_scale)
        self.net.reduce_grads(scale=scale)
        if not getattr(self._module, '_learn_std', True):
            self.net.grads[0:1].zero_()
        if self.grad_hook is not None:
            self.grad_hook(self.net.grads)
        self.apply_step()
