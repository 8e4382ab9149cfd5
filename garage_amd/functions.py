"""Device versions of garage's free functions on the hot path.

``discount_cumsum`` (``np/_functions.py:111-128``), ``compute_advantages`` and
``filter_valids`` (``torch/_functions.py:25-85,119-132``), ``pad_batch_array``
(``np/_functions.py:375-406``): same names, argument order and results, computed
by the HIP scan kernel.  ``log_performance`` / ``log_multitask_performance``
(``_functions.py:177-275``): the per-episode reductions run on the device, the
grouping by task and the tabular rows follow the reference.
"""
import numpy as np
import torch

from garage_amd import logger
from garage_amd._dtypes import StepType, step_types_as_uint8
from garage_amd._dtypes import pad_batch_array  # noqa: F401  (host helper)
from garage_amd._lib import call, dptr, stream_ptr
from garage_amd.engine import gae_scan, require_gpu


def discount_cumsum(x, discount):
    """``y[t] = x[t] + discount * y[t+1]`` along the last axis.

    numpy in -> numpy float64 out (like the reference); device tensor in ->
    device fp32 tensor out.  The recurrence runs in fp64 registers on the GPU.
    """
    dev = require_gpu()
    is_np = not torch.is_tensor(x)
    t = torch.as_tensor(np.asarray(x, dtype=np.float32) if is_np else x)
    t = t.to(dev, torch.float32)
    shape = t.shape
    rows = t.reshape(-1, shape[-1]).contiguous()
    zeros = torch.zeros_like(rows)
    # lambda = 1, values = 0: the "advantage" slot is unused, returns = scan
    _, ret = gae_scan(rows, zeros, discount=discount, gae_lambda=1.0,
                      max_episode_length=rows.shape[1])
    ret = ret.reshape(shape)
    return ret.cpu().numpy().astype(np.float64) if is_np else ret


def compute_advantages(discount, gae_lambda, max_episode_length, baselines,
                       rewards):
    """GAE over zero padded ``(N, P)`` tensors (``torch/_functions.py:25-85``).

    Whatever sits in the padding of ``baselines`` takes part, exactly as in the
    reference's convolution.
    """
    dev = require_gpu()
    b = torch.as_tensor(baselines).to(dev, torch.float32).contiguous()
    r = torch.as_tensor(rewards).to(dev, torch.float32).contiguous()
    if r.shape[1] != max_episode_length:
        raise ValueError('rewards must have max_episode_length columns')
    adv, _ = gae_scan(r, b, discount=discount, gae_lambda=gae_lambda,
                      max_episode_length=max_episode_length)
    return adv


def filter_valids(tensor, valids):
    """``torch/_functions.py:119-132``."""
    return [tensor[i][:int(v)] for i, v in enumerate(valids)]


# -- log_performance / log_multitask_performance (_functions.py:177-275) --------
_STAT_KEYS = ('AverageDiscountedReturn', 'AverageReturn', 'StdReturn',
              'MaxReturn', 'MinReturn', 'TerminationRate', 'SuccessRate')


def episode_statistics(batch, discount, returns=None):
    """Per-episode numbers ``log_performance`` reduces, computed on the device.

    Returns ``(undiscounted[N] f64, first_discounted_return[N] f64,
    terminated[N] f64, success[N] f64 or None)``: the episode's reward sum, the
    first element of its ``discount_cumsum`` (``returns``: packed discounted
    returns if the caller already has them), whether any step is ``TERMINAL``
    and whether any step reported ``env_infos['success']``.
    """
    dev = require_gpu()
    lengths = np.asarray(batch.lengths, dtype=np.int64)
    N = int(lengths.shape[0])
    if hasattr(batch, 'rewards_dev'):
        rew, st, off = batch.rewards_dev, batch.step_types_dev, batch.ep_off_dev
    else:
        rew = torch.from_numpy(
            np.ascontiguousarray(batch.rewards, dtype=np.float32)).to(dev)
        st = torch.from_numpy(step_types_as_uint8(batch.step_types)).to(dev)
        off = torch.from_numpy(
            np.concatenate([[0], np.cumsum(lengths)])).to(dev)
    sums = torch.empty(N, dtype=torch.float64, device=dev)
    call('ga_episode_sums_f32', dptr(rew), dptr(off), N, dptr(sums),
         stream_ptr())
    if returns is None:
        # lambda = 1, values = 0: the scan's return slot is discount_cumsum
        _, returns = gae_scan(rew, torch.zeros_like(rew), discount=discount,
                              gae_lambda=1.0,
                              max_episode_length=int(lengths.max()),
                              offsets=off, max_len=int(lengths.max()))
    first = returns[off[:-1]].to(torch.float64)
    # an episode ends at its first TERMINAL / TIMEOUT step, so only the last
    # step of a complete episode can be TERMINAL; fragments may hold none
    is_term = (st == int(StepType.TERMINAL)).to(torch.float32)
    term = torch.empty(N, dtype=torch.float64, device=dev)
    call('ga_episode_sums_f32', dptr(is_term), dptr(off), N, dptr(term),
         stream_ptr())
    term = (term > 0).to(torch.float64)
    host = torch.cat([sums, first, term]).cpu().numpy()
    success = None
    if 'success' in batch.env_infos:
        flags = np.asarray(batch.env_infos['success']).reshape(
            int(lengths.sum()), -1).any(axis=1)
        starts = np.concatenate([[0], np.cumsum(lengths)[:-1]])
        success = np.logical_or.reduceat(flags, starts).astype(np.float64)
    return host[:N], host[N:2 * N], host[2 * N:], success


def _record_rows(itr, prefix, und, disc, term, success):
    tab = logger.tabular
    with tab.prefix(prefix + '/'):
        tab.record('Iteration', itr)
        tab.record('NumEpisodes', len(und))
        tab.record('AverageDiscountedReturn', np.mean(disc))
        tab.record('AverageReturn', np.mean(und))
        tab.record('StdReturn', np.std(und))
        tab.record('MaxReturn', np.max(und))
        tab.record('MinReturn', np.min(und))
        tab.record('TerminationRate', np.mean(term))
        if success is not None:
            tab.record('SuccessRate', np.mean(success))


def log_performance(itr, batch, discount, prefix='Evaluation', returns=None):
    """``_functions.py:233-275``: records ``<prefix>/{Iteration, NumEpisodes,
    AverageDiscountedReturn, AverageReturn, StdReturn, MaxReturn, MinReturn,
    TerminationRate[, SuccessRate]}`` and returns the undiscounted returns."""
    und, disc, term, success = episode_statistics(batch, discount, returns)
    _record_rows(itr, prefix, und, disc, term, success)
    return list(und)


def log_multitask_performance(itr, batch, discount, name_map=None):
    """``_functions.py:177-230``: one block of rows per task plus ``Average/``.

    Episodes are grouped by the ``task_name`` env-info of their first step,
    else by ``name_map[task_id]`` (``'Task #<id>'`` when unmapped), else under
    ``'__unnamed_task__'``.  With a ``name_map`` exactly its tasks are logged,
    absent ones as NaN rows with ``NumEpisodes = 0``.  As in the reference
    (``:204``) task ids WITHOUT a ``name_map`` produce no per-task rows: the map
    is replaced by an empty dict, whose values are then the tasks to log.
    """
    und, disc, term, success = episode_statistics(batch, discount)
    lengths = np.asarray(batch.lengths, dtype=np.int64)
    starts = np.concatenate([[0], np.cumsum(lengths)[:-1]])
    groups = {}
    for e, start in enumerate(starts):
        name = '__unnamed_task__'
        if 'task_name' in batch.env_infos:
            name = batch.env_infos['task_name'][start]
        elif 'task_id' in batch.env_infos:
            name_map = {} if name_map is None else name_map
            task_id = batch.env_infos['task_id'][start]
            name = name_map.get(task_id, 'Task #{}'.format(task_id))
        groups.setdefault(name, []).append(e)
    names = list(groups) if name_map is None else list(name_map.values())
    tab = logger.tabular
    for name in names:
        if name in groups:
            sel = np.asarray(groups[name])
            _record_rows(itr, name, und[sel], disc[sel], term[sel],
                         None if success is None else success[sel])
        else:
            with tab.prefix(name + '/'):
                tab.record('Iteration', itr)
                tab.record('NumEpisodes', 0)
                for k in _STAT_KEYS:
                    tab.record(k, np.nan)
    _record_rows(itr, 'Average', und, disc, term, success)
    return list(und)
