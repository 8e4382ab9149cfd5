"""Developer tool: random network shapes / options / minibatch sizes, one PPO or VPG
iteration each, HIP path against the oracle (parameters and logged scalars).

    python tools/fuzz_update.py [n_cases] [first_seed]

A BAD case is not necessarily a bug: 24 fp32 Adam steps on PPO's clipped objective
are discontinuous in the inputs (a sample whose ratio sits on the clip boundary, a
gradient element at rounding-noise level that Adam turns into a full step).  To
tell: FUZZ_ORACLE_THREADS=1 / =8 change the summation order of the ORACLE's own
CPU GEMMs, FUZZ_DUMP=file.npz saves both sides' parameters -- if the oracle
differs from itself by as much as the HIP path differs from it, the case is at the
noise floor of the problem, not of the kernels.  FUZZ_EPOCHS / FUZZ_MB re-run a case
with fewer steps: a kernel bug shows in the first steps, a boundary event only
once the ratios have moved (seed 1069 of the default range: 3e-8 after the first
epoch's 12 steps, 8e-4 after the second's; seed 5085 likewise).
"""
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from garage_amd._dtypes import (Box, Discrete, EnvSpec, EpisodeBatch,  # noqa: E402
                                StepType)
from garage_amd.algos import PPO, VPG  # noqa: E402
from garage_amd.optimizers import OptimizerWrapper  # noqa: E402
from garage_amd.policies import (CategoricalMLPPolicy,  # noqa: E402
                                 GaussianMLPPolicy, GaussianMLPValueFunction)
from oracle import batch as ob  # noqa: E402
from oracle import networks as nets  # noqa: E402
from oracle.ppo import OraclePPO  # noqa: E402

WIDTHS = [5, 8, 17, 32, 48, 64, 96, 100, 128, 160, 256, 300]
ACTS = {'tanh': torch.tanh, 'relu': torch.relu, 'none': None}


def one_case(seed):
    rng = np.random.RandomState(seed)
    O = int(rng.choice([1, 3, 4, 9, 17, 24, 33, 40]))
    A = int(rng.choice([1, 2, 3, 6, 8, 9, 17]))
    nh = int(rng.choice([1, 2, 2, 2, 3]))
    square = rng.rand() < 0.5
    w0 = int(rng.choice(WIDTHS))
    hidden = tuple(w0 if square else int(rng.choice(WIDTHS)) for _ in range(nh))
    P = int(rng.choice([8, 16, 40]))
    n_eps = int(rng.choice([30, 80, 200]))
    lens = rng.randint(1, P + 1, size=n_eps)
    lens[0] = P
    S = int(lens.sum())
    mb = int(rng.choice([0, 17, 64, 97, 256, 1000]))
    mb = None if mb == 0 else mb
    # long fp32 optimisation trajectories diverge on their own: at most ~12
    # minibatches per pass (two passes), small steps
    if mb is not None and S > 12 * mb:
        mb = -(-S // 12)
    pa = str(rng.choice(['tanh', 'tanh', 'tanh', 'relu', 'none']))
    va = str(rng.choice(['tanh', 'tanh', 'relu']))
    # (LayerNorm over ONE feature is excluded: x - mean is exactly 0 there, the
    # reference's backward kernel returns rounding noise of ~1e-5 for d(gamma) where
    # this one returns 0, and Adam turns any non-zero gradient into a full step)
    ln = bool(rng.rand() < 0.2) and O > 1
    discrete = bool(rng.rand() < 0.2)
    softplus = bool(rng.rand() < 0.25)
    out_tanh = bool(rng.rand() < 0.15)
    algo_name = str(rng.choice(['ppo', 'ppo', 'vpg']))
    kw = {}
    ent = rng.rand()
    if ent < 0.25:
        kw = dict(entropy_method='regularized', policy_ent_coeff=0.02)
    elif ent < 0.4:
        kw = dict(entropy_method='max', policy_ent_coeff=0.02, center_adv=False,
                  stop_entropy_gradient=True)
    if discrete:
        A = max(A, 2)
        softplus = out_tanh = False
    desc = dict(seed=seed, O=O, A=A, hidden=hidden, S=S, mb=mb, pa=pa, va=va, ln=ln,
                softplus=softplus, out_tanh=out_tanh, algo=algo_name,
                discrete=discrete, **kw)
    spec = EnvSpec(Box(-np.inf, np.inf, (O, )),
                   Discrete(A) if discrete else Box(-np.inf, np.inf, (A, )),
                   max_episode_length=P)
    torch.manual_seed(seed)
    pkw = dict(hidden_nonlinearity=ACTS[pa], layer_normalization=ln)
    if softplus:
        pkw.update(std_parameterization='softplus', init_std=0.7)
    if out_tanh:
        pkw.update(output_nonlinearity=torch.tanh)
    if discrete:
        pol = CategoricalMLPPolicy(spec, hidden_sizes=hidden,
                                   hidden_nonlinearity=ACTS[pa],
                                   layer_normalization=ln)
    else:
        pol = GaussianMLPPolicy(spec, hidden_sizes=hidden, **pkw)
    vf = GaussianMLPValueFunction(spec, hidden_sizes=hidden,
                                  hidden_nonlinearity=ACTS[va])
    st = []
    for L in lens:
        t = [1] * L
        t[0] = 0
        t[-1] = 3 if L == P else 2
        st += t
    obs = rng.randn(S, O).astype(np.float32)
    acts = (rng.randint(0, A, size=S).astype(np.int64) if discrete else
            (0.7 * rng.randn(S, A)).astype(np.float32))
    rew = rng.randn(S)
    E = int(os.environ.get('FUZZ_EPOCHS', 2))  # (FUZZ_EPOCHS / FUZZ_MB: bisect a BAD case)
    if os.environ.get('FUZZ_MB'):
        mb = int(os.environ['FUZZ_MB']) or None
    with nets.hidden_nonlinearity(policy=ACTS[pa], value=ACTS[va]), \
            nets.output_nonlinearity(policy=torch.tanh if out_tanh else None), \
            nets.std_parameterization('softplus' if softplus else 'exp'):
        oracle = OraclePPO(OrderedDict(pol.state_dict()),
                           OrderedDict(vf.state_dict()), max_episode_length=P,
                           algo=algo_name,
                           policy_kind='categorical' if discrete else 'gaussian',
                           max_optimization_epochs=E,
                           minibatch_size=mb, policy_lr=3e-4, vf_lr=3e-4,
                           gae_lambda=0.95, **kw)
        b = ob.OracleEpisodeBatch(
            observations=obs, last_observations=np.zeros((len(lens), O),
                                                         np.float32),
            actions=acts, rewards=rew, step_types=np.asarray(st), lengths=lens,
            max_episode_length=P)
        np.random.seed(seed)
        want = oracle.train_once(b)
        wpol, wvf = oracle.state()
    cls = PPO if algo_name == 'ppo' else VPG
    algo = cls(env_spec=spec, policy=pol, value_function=vf, sampler=None,
               policy_optimizer=OptimizerWrapper(
                   (torch.optim.Adam, dict(lr=3e-4)), pol,
                   max_optimization_epochs=E, minibatch_size=mb),
               vf_optimizer=OptimizerWrapper(
                   (torch.optim.Adam, dict(lr=3e-4)), vf,
                   max_optimization_epochs=E, minibatch_size=mb),
               gae_lambda=0.95, **kw)
    batch = EpisodeBatch(env_spec=spec, episode_infos={}, observations=obs,
                         last_observations=np.zeros((len(lens), O), np.float32),
                         actions=acts, rewards=rew, env_infos={}, agent_infos={},
                         step_types=np.asarray([StepType(s) for s in st],
                                               dtype=object),
                         lengths=lens.astype('l'))
    np.random.seed(seed)
    algo._train_once(0, batch)
    worst = 0.0
    bad = []
    for k in ('policy/LossBefore', 'policy/LossAfter', 'policy/KL', 'policy/Entropy',
              'vf/LossBefore', 'vf/LossAfter'):
        d = abs(algo.last_tabular[k] - want[k])
        tol = 3e-5 + 3e-4 * abs(want[k])
        if not d <= tol:
            bad.append((k, algo.last_tabular[k], want[k]))
    for mine, theirs in ((pol.state_dict(), wpol), (vf.state_dict(), wvf)):
        for k, v in mine.items():
            d = np.abs(v.numpy() - np.asarray(theirs[k]))
            worst = max(worst, float(d.max()))
            if not (d.max() <= 2e-4 and d.mean() <= 5e-6):
                bad.append((k, float(d.max()), float(d.mean())))
    dump = os.environ.get('FUZZ_DUMP')
    if dump:  # oracle and HIP parameters of the case, for a closer look at a BAD one
        np.savez(dump,
                 **{'oracle/' + k: np.asarray(v) for d_ in (wpol, wvf)
                    for k, v in d_.items()},
                 **{'hip/' + k: v.numpy() for d_ in (pol.state_dict(), vf.state_dict())
                    for k, v in d_.items()})
    return desc, worst, bad


def main():
    if os.environ.get('FUZZ_ORACLE_THREADS'):
        torch.set_num_threads(int(os.environ['FUZZ_ORACLE_THREADS']))
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    failures = 0
    for seed in range(first, first + n):
        try:
            desc, worst, bad = one_case(seed)
        except Exception as exc:  # noqa: BLE001 - report and go on
            print('CASE', seed, 'RAISED', type(exc).__name__, exc, flush=True)
            failures += 1
            continue
        tag = 'ok ' if not bad else 'BAD'
        print(tag, desc, 'worst param diff %.2e' % worst, bad[:3], flush=True)
        failures += bool(bad)
    print('failures:', failures, 'of', n)
    return 1 if failures else 0


if __name__ == '__main__':
    sys.exit(main())
