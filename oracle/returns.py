"""Oracle: discounted returns, GAE(lambda) advantages, padding helpers.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Restates
  * ``np/_functions.py:111-128``   discount_cumsum (scipy lfilter, float64)
  * ``np/_functions.py:375-406``   pad_batch_array
  * ``torch/_functions.py:25-85``  compute_advantages (zero-padded fp32
                                   correlation with [1, c, c^2, ...])
  * ``torch/_functions.py:119-132`` filter_valids
  * ``torch/algos/vpg.py:349-379`` VPG._compute_advantage (centre / shift)
and adds float64 recursions of the same quantities that scale to the full
benchmark sizes (used for size-independent property tests).
"""
import warnings

import numpy as np
import scipy.signal
import torch
import torch.nn.functional as F


def discount_cumsum(x, discount):
    """y[t] = x[t] + discount * y[t+1] along the last axis.

    The reference (``np/_functions.py:127``) runs the IIR filter
    ``1 / (1 - discount z^-1)`` over the time-reversed signal with
    ``scipy.signal.lfilter`` (scipy is unpinned in the reference's
    ``setup.py:23``; 1.15.3 here).  Same call, so same float64 rounding.
    """
    x = np.asarray(x)
    flipped = x[..., ::-1]
    y = scipy.signal.lfilter([1.0], [1.0, -float(discount)], flipped, axis=-1)
    return y[..., ::-1]


def discount_cumsum_recursive(x, discount):
    """Plain float64 recursion of the same recurrence (no scipy).

    Used to show the lfilter call is nothing more than the recurrence, and as
    the reference for the HIP scan at sizes where only a vectorised loop over
    time is affordable.
    """
    x = np.asarray(x, dtype=np.float64)
    out = np.empty_like(x)
    acc = np.zeros(x.shape[:-1], dtype=np.float64)
    for t in range(x.shape[-1] - 1, -1, -1):
        acc = x[..., t] + discount * acc
        out[..., t] = acc
    return out


def pad_batch_array(array, lengths, max_length=None):
    """Packed ``(sum(lengths), X*)`` -> zero padded ``(N, max_length, X*)``.

    ``np/_functions.py:375-406``: asserts the packed size, widens (with a
    warning) when an episode is longer than ``max_length``.
    """
    lengths = [int(v) for v in lengths]
    assert array.shape[0] == sum(lengths)
    longest = max(lengths)
    if max_length is None:
        max_length = longest
    elif max_length < longest:
        warnings.warn('Creating a padded array with longer length than '
                      'requested')
        max_length = longest
    out = np.zeros((len(lengths), max_length) + array.shape[1:],
                   dtype=array.dtype)
    cursor = 0
    for row, n in enumerate(lengths):
        out[row, :n] = array[cursor:cursor + n]
        cursor += n
    return out


def filter_valids(tensor, valids):
    """``torch/_functions.py:119-132``: the first ``valids[i]`` items of row i."""
    return [tensor[i][:int(v)] for i, v in enumerate(valids)]


def compute_advantages(discount, gae_lambda, max_episode_length, baselines,
                       rewards):
    """GAE over zero-padded ``(N, P)`` fp32 tensors, as the reference does it.

    ``torch/_functions.py:75-84``: ``delta_t = r_t + g*b_{t+1} - b_t`` with
    ``b_P := 0``; ``A_t = sum_k (g*l)^k delta_{t+k}`` evaluated as a valid
    cross-correlation of the right-zero-padded deltas with the length-P kernel
    ``[1, c, c^2, ...]`` (``F.conv2d`` there; ``F.conv1d`` here -- the same
    fp32 MAC sequence per output on the CPU backend).

    NB: ``baselines`` of short episodes are *not* zero in the
    padding when they come from ``value_function(padded_obs)``; whatever sits
    there takes part in the sum.  That is the reference's behaviour and it is
    reproduced, not fixed.
    """
    P = int(max_episode_length)
    c = discount * gae_lambda
    kernel = torch.full((P - 1, ), c, dtype=torch.float)
    kernel = torch.cumprod(F.pad(kernel, (1, 0), value=1.0), dim=0)
    next_b = F.pad(baselines, (0, 1))[:, 1:]
    deltas = rewards + discount * next_b - baselines
    padded = F.pad(deltas, (0, P - 1)).unsqueeze(1)  # (N, 1, 2P-1)
    adv = F.conv1d(padded, kernel.view(1, 1, P))
    return adv.reshape(rewards.shape)


def gae_padded_f64(discount, gae_lambda, baselines, rewards):
    """Float64 backward recursion over the *whole* padded row.

    Mathematically identical to :func:`compute_advantages` (no special casing
    of short episodes: the padding columns take part exactly as they do in the
    reference's convolution).
    """
    b = np.asarray(baselines, dtype=np.float64)
    r = np.asarray(rewards, dtype=np.float64)
    N, P = r.shape
    adv = np.empty((N, P), dtype=np.float64)
    carry = np.zeros(N, dtype=np.float64)
    nxt = np.zeros(N, dtype=np.float64)
    c = discount * gae_lambda
    for t in range(P - 1, -1, -1):
        delta = r[:, t] + discount * nxt - b[:, t]
        carry = delta + c * carry
        adv[:, t] = carry
        nxt = b[:, t]
    return adv


def gae_ragged_closed_form_f64(discount, gae_lambda, max_episode_length,
                               values, rewards, lengths, v0):
    """Same numbers as the reference for ragged episodes, without the padding.

    ``values``/``rewards`` are packed ``(S,)``; ``v0`` is what the value
    function returns for an all-zero observation (the content of every padded
    baseline cell, ``vpg.py:147,155-156``).  For an episode of length ``L < P``
    the padded tail contributes an initial carry::

        m = P - L, c = g*l
        C0 = (g-1)*v0*(1 - c^(m-1))/(1-c) - c^(m-1)*v0        (c != 1)
        C0 = (g-1)*v0*(m-1) - v0                              (c == 1)

    and the last valid step bootstraps from ``v0``; for ``L == P`` the carry
    and the bootstrap are 0.  This is the recipe the HIP scan implements.
    """
    values = np.asarray(values, dtype=np.float64)
    rewards = np.asarray(rewards, dtype=np.float64)
    P = int(max_episode_length)
    c = discount * gae_lambda
    out = np.empty_like(values)
    start = 0
    for L in (int(v) for v in lengths):
        m = P - L
        if m <= 0:
            carry, boot = 0.0, 0.0
        else:
            if c == 1.0:
                geo = float(m - 1)
            else:
                geo = (1.0 - c**(m - 1)) / (1.0 - c)
            carry = (discount - 1.0) * v0 * geo - c**(m - 1) * v0
            boot = v0
        nxt = boot
        for t in range(start + L - 1, start - 1, -1):
            carry = rewards[t] + discount * nxt - values[t] + c * carry
            out[t] = carry
            nxt = values[t]
        start += L
    return out


def vpg_compute_advantage(discount, gae_lambda, max_episode_length, rewards,
                          valids, baselines, center_adv=True,
                          positive_adv=False):
    """``VPG._compute_advantage`` (``torch/algos/vpg.py:349-379``).

    Packs the padded advantages, then ``(a - mean) / (var + 1e-8)`` with the
    *unbiased* variance (yes, variance -- SURVEY.md Q1), then the optional
    shift by the minimum.
    """
    adv = compute_advantages(discount, gae_lambda, max_episode_length,
                             baselines, rewards)
    flat = torch.cat(filter_valids(adv, valids))
    if center_adv:
        flat = (flat - flat.mean()) / (flat.var() + 1e-8)
    if positive_adv:
        flat = flat - flat.min()
    return flat


def padded_returns(padded_rewards, discount):
    """``vpg.py:149-153``: row-wise float64 lfilter, then a cast to fp32."""
    rows = [discount_cumsum(row, discount) for row in padded_rewards]
    return torch.Tensor(np.stack(rows))
