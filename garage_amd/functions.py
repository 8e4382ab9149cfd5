"""Device versions of garage's free functions on the hot path.

``discount_cumsum`` (``np/_functions.py:111-128``), ``compute_advantages`` and
``filter_valids`` (``torch/_functions.py:25-85,119-132``), ``pad_batch_array``
(``np/_functions.py:375-406``): same names, argument order and results, computed
by the HIP scan kernel.
"""
import numpy as np
import torch

from garage_amd._dtypes import pad_batch_array  # noqa: F401  (host helper)
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
