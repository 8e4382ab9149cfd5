"""Developer tool: rounding error of ONE optimizer step's gradients against an fp64
reference, for the exact fp32 kernels and for the opt-in split-operand (3 x bf16)
k-loops (``ga_set_split_bf16``), on a C3-shaped problem (obs 17, act 6, MLP(256, 256),
one minibatch of 64 x 256 = 16384 rows).  The reference is the oracle's loss
(``oracle/ppo.py``, ``oracle/networks.py``) evaluated in float64 on the SAME
parameters, observations, actions, advantages and returns, differentiated by autograd.

    python tools/split_error_histogram.py [--envs 64]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench  # noqa: E402
from garage_amd import _lib  # noqa: E402
import test_configs_gpu as TC  # noqa: E402
from oracle import networks as nets  # noqa: E402
from oracle.ppo import OraclePPO  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=64)
    args = ap.parse_args()
    lib = _lib.load()
    cfg = bench.CONFIGS['c3']
    n, T = args.envs, cfg['T']
    S = n * T
    got = {}
    ref = None
    try:
        for mode in ('exact', 'split'):
            lib.ga_set_split_bf16(1 if mode == 'split' else 0)
            algo, sampler, pol, vf = TC._build(cfg, n, 1, S)
            sd_p, sd_v = pol.state_dict(), vf.state_dict()
            eps = sampler.obtain_samples(0, S, None)
            np.random.seed(40)
            algo._train_once(0, eps)
            g = {}
            for name, mod in (('policy', pol), ('vf', vf)):
                for key, view in mod.net.named_views(mod.net.grads):
                    g[name + '/' + key] = view.detach().cpu().double().clone()
            got[mode] = g
            if ref is None:
                adv = algo.last_tensors['advantages'].cpu().double()
                ret = algo.last_tensors['returns'].cpu().double()
                obs = torch.from_numpy(np.asarray(eps.observations)).double()
                act = torch.from_numpy(np.asarray(eps.actions)).double()
                o = OraclePPO({k: v.double() for k, v in sd_p.items()},
                              {k: v.double() for k, v in sd_v.items()},
                              max_episode_length=T, max_optimization_epochs=1,
                              minibatch_size=S)
                o._policy_loss(obs, act, adv).backward()
                nets.value_loss(o.value, obs, ret).backward()
                ref = {}
                for name, params, mod in (('policy', o.policy, pol), ('vf', o.value, vf)):
                    for key, _ in mod.net.named_views():
                        full = [k for k in params if k.endswith(
                            'init_std' if key == '_init_std' else key)]
                        grad = params[full[0]].grad
                        ref[name + '/' + key] = (torch.zeros(1, dtype=torch.float64)
                                                 if grad is None else grad.clone())
    finally:
        lib.ga_set_split_bf16(0)
    edges = [0.0] + [10.0 ** e for e in range(-10, -3)] + [np.inf]
    out = {'rows': S, 'tensors': {}}
    for key in ref:
        r = ref[key].reshape(-1)
        scale = float(r.abs().max())
        if scale == 0.0:
            continue
        row = {'max_abs_fp64': scale}
        for mode in ('exact', 'split'):
            e = (got[mode][key].reshape(-1) - r).abs() / scale
            hist = np.histogram(e.numpy(), bins=edges)[0].tolist()
            row[mode] = {'max_rel_to_max': float(e.max()), 'rms_rel_to_max':
                         float((e ** 2).mean().sqrt()), 'hist': hist}
        row['split_minus_exact_max_rel'] = float(
            (got['split'][key] - got['exact'][key]).abs().max() / scale)
        out['tensors'][key] = row
    out['hist_edges_rel_to_max'] = [str(x) for x in edges]
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
