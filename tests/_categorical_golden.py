"""Shared by the CPU (oracle) and GPU tests of ``train_once_categorical.npz``.

The fixture was recorded from the reference's own torch categorical policy,
``CategoricalCNNPolicy`` (``torch/policies/categorical_cnn_policy.py``), built
over a ``(O, 1, 1)`` observation with one 1 x 1 convolution of ``hidden[0]``
channels and its MLP with the remaining hidden sizes -- a 1 x 1 convolution of
a 1 x 1 image is a dense layer, so that class IS a tanh MLP(hidden) with the
reference's ``Categorical(logits=softmax(scores))`` head (``:138-139``) --
trained by the real ``PPO`` / ``VPG._train_once`` (``tests/golden/
make_golden.py:gen_train_once_categorical``).  This module only renames its
parameters to the MLP naming the oracle and ``garage_amd`` use.
"""
from collections import OrderedDict

import numpy as np
import torch

CASES = {
    'ppo': dict(),
    'ppo_reg': dict(entropy_method='regularized', policy_ent_coeff=0.02),
    'ppo_pos3': dict(positive_adv=True),
    'vpg': dict(),
    'ppo_full': dict(),
    'ppo_c2': dict(),
}

LOG_KEYS = {
    'policy/LossBefore': 'CategoricalCNNPolicy/LossBefore',
    'policy/LossAfter': 'CategoricalCNNPolicy/LossAfter',
    'policy/dLoss': 'CategoricalCNNPolicy/dLoss',
    'policy/KLBefore': 'CategoricalCNNPolicy/KLBefore',
    'policy/KL': 'CategoricalCNNPolicy/KL',
    'policy/Entropy': 'CategoricalCNNPolicy/Entropy',
    'vf/LossBefore': 'GaussianMLPValueFunction/LossBefore',
    'vf/LossAfter': 'GaussianMLPValueFunction/LossAfter',
    'vf/dLoss': 'GaussianMLPValueFunction/dLoss',
}

PREFIX = '_module._mean_module.'


def mlp_name(ref_name):
    """``CategoricalCNNPolicy`` parameter name -> the MLP module's name."""
    if ref_name.startswith('_cnn_module._cnn_layers.conv_0.'):
        return PREFIX + '_layers.0.linear.' + ref_name.rsplit('.', 1)[1]
    assert ref_name.startswith('_mlp_module.'), ref_name
    rest = ref_name[len('_mlp_module.'):]
    if rest.startswith('_layers.'):
        _, i, tail = rest.split('.', 2)
        return PREFIX + '_layers.{}.{}'.format(int(i) + 1, tail)
    return PREFIX + rest  # _output_layers.0.linear.*


def policy_params(g, prefix):
    """Policy parameters stored under ``prefix`` in MLP naming (conv weights
    ``(H, O, 1, 1)`` as ``(H, O)`` matrices)."""
    out = OrderedDict()
    for k in g.files:
        if k.startswith(prefix):
            v = g[k]
            if v.ndim == 4:
                v = v.reshape(v.shape[0], v.shape[1])
            out[mlp_name(k[len(prefix):])] = torch.from_numpy(v.copy())
    return out


def value_params(g, prefix):
    out = OrderedDict()
    for k in g.files:
        if k.startswith(prefix):
            out[k[len(prefix):]] = torch.from_numpy(g[k].copy())
    return out


def flat(a):
    a = np.asarray(a)
    return a.reshape(a.shape[0], -1) if a.ndim == 4 else a
