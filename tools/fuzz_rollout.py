"""Developer tool: random rollout shapes (envs, observation / action widths, episode
length ranges, network shapes and activations, sample counts over several
``obtain_samples`` calls), ``GpuVecSampler`` + ``SyntheticVecEnv`` against the oracle's
``VecWorker`` stepping the per-env CPU twins, teacher-forced with the same noise:
bookkeeping (lengths, step types, episode order, observations) must be exact.

    python tools/fuzz_rollout.py [n_cases] [first_seed]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from garage_amd.envs import SyntheticVecEnv  # noqa: E402
from garage_amd.policies import GaussianMLPPolicy  # noqa: E402
from garage_amd.sampler import GpuVecSampler, GpuVecWorker  # noqa: E402
from oracle import envs as oenvs  # noqa: E402
from oracle import networks as nets  # noqa: E402
from oracle import sampler as osamp  # noqa: E402

ACTS = {'tanh': torch.tanh, 'relu': torch.relu}


def one_case(seed):
    rng = np.random.RandomState(seed)
    n = int(rng.choice([1, 2, 7, 33, 64, 100, 257]))
    O = int(rng.choice([1, 3, 5, 17, 32, 40]))
    A = int(rng.choice([1, 2, 3, 6, 8]))
    P = int(rng.choice([2, 5, 12, 31]))
    min_len = None if rng.rand() < 0.4 else int(rng.randint(1, P + 1))
    nh = int(rng.choice([1, 2, 2, 3]))
    hidden = tuple(int(rng.choice([4, 16, 32, 64, 100, 256, 300])) for _ in range(nh))
    act = str(rng.choice(['tanh', 'tanh', 'relu']))
    calls = [int(rng.randint(1, 3 * n * P + 2)) for _ in range(int(rng.randint(1, 4)))]
    desc = dict(seed=seed, n=n, O=O, A=A, P=P, min_len=min_len, hidden=hidden,
                act=act, calls=calls)
    torch.manual_seed(seed)
    env = SyntheticVecEnv(n, O, A, P, min_len=min_len, seed=seed)
    pol = GaussianMLPPolicy(env.spec, hidden_sizes=hidden,
                            hidden_nonlinearity=ACTS[act])
    dev = pol.device
    n_noise = sum(-(-c // n) for c in calls) + 4 * P + 8
    noise = torch.randn(n_noise, n, 8)

    def noise_fn(step):
        return noise[step].to(dev)

    sampler = GpuVecSampler(pol, env, max_episode_length=P, n_workers=1,
                            worker_class=GpuVecWorker,
                            worker_args=dict(n_envs=n, noise_fn=noise_fn))
    params = pol.state_dict()

    class CpuPolicy:
        calls = 0

        def reset(self, do_resets=None):
            pass

        def get_actions(self, obs):
            with torch.no_grad(), nets.hidden_nonlinearity(policy=ACTS[act]):
                dist, info = nets.policy_forward(
                    params, torch.from_numpy(np.asarray(obs, np.float32)))
            a = dist.mean + dist.stddev * noise[self.calls][:, :A]
            self.calls += 1
            return a.numpy(), {'mean': info['mean'].numpy()}

    ref = osamp.OracleLocalSampler(
        CpuPolicy(),
        [[oenvs.SyntheticEnv(i, O, A, P, min_len=min_len, seed=seed)
          for i in range(n)]], max_episode_length=P, n_workers=1,
        worker_class=osamp.OracleVecWorker, worker_args=dict(n_envs=n))
    bad = []
    for k, num in enumerate(calls):
        eps = sampler.obtain_samples(k, num, None)
        want = ref.obtain_samples(k, num, None)
        if not np.array_equal(eps.lengths, want.lengths):
            bad.append(('lengths', k))
            break
        if not np.array_equal([int(s) for s in eps.step_types],
                              [int(s) for s in want.step_types]):
            bad.append(('step_types', k))
        if not np.array_equal(eps.observations, want.observations):
            bad.append(('observations', k))
        if not np.array_equal(eps.last_observations, want.last_observations):
            bad.append(('last_observations', k))
        if not np.allclose(eps.actions, want.actions, atol=2e-5):
            bad.append(('actions', k, float(np.abs(eps.actions - want.actions).max())))
        if not np.allclose(eps.rewards, want.rewards, atol=2e-5):
            bad.append(('rewards', k))
        if not np.allclose(eps.agent_infos['mean'], want.agent_infos['mean'],
                           atol=2e-5):
            bad.append(('mean', k))
    if sampler.total_env_steps != ref.total_env_steps:
        bad.append(('total_env_steps', sampler.total_env_steps, ref.total_env_steps))
    return desc, bad


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    failures = 0
    for seed in range(first, first + n):
        try:
            desc, bad = one_case(seed)
        except Exception as exc:  # noqa: BLE001 - report and go on
            print('CASE', seed, 'RAISED', type(exc).__name__, exc, flush=True)
            failures += 1
            continue
        print('ok ' if not bad else 'BAD', desc, bad[:4], flush=True)
        failures += bool(bad)
    print('failures:', failures, 'of', n)
    return 1 if failures else 0


if __name__ == '__main__':
    sys.exit(main())
