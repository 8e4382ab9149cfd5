"""GpuFragmentWorker against the real ``FragmentWorker`` golden
(tests/golden/sampler.npz, ``frag1_`` / ``frag2_``)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('tpc', [1, 2])
def test_fragments_match_real_fragment_worker(golden, tpc):
    from garage_amd._dtypes import Box, EnvSpec
    from garage_amd.policies import GaussianMLPPolicy
    from garage_amd.sampler import GpuFragmentWorker, GpuVecSampler
    from oracle import envs as oenvs
    g = golden('sampler')
    P, n = [int(v) for v in g['cfg']]
    cyc = g['cycles']
    spec = EnvSpec(Box(-np.inf, np.inf, (3, )), Box(-np.inf, np.inf, (2, )),
                   max_episode_length=P)

    class Env(oenvs.CountingEnv):

        def __init__(self, i):
            super().__init__(i, cyc[i], P)
            self.spec = spec

    pol = GaussianMLPPolicy(spec, hidden_sizes=(), init_std=1.0)
    pol.net.weight(0).copy_(torch.tensor([[1., 1., 1.], [0., 0., 0.]]))
    pol.net.bias(0).zero_()
    dev = pol.device

    def noise_fn(step):
        z = torch.zeros(n, 4, device=dev)
        z[:, 1] = float(step)
        return z

    sampler = GpuVecSampler(
        pol, [[Env(i) for i in range(n)]], max_episode_length=P, n_workers=1,
        worker_class=GpuFragmentWorker,
        worker_args=dict(n_envs=n, timesteps_per_call=tpc, noise_fn=noise_fn))
    eps = sampler.obtain_samples(0, 20, None)
    pre = 'frag%d_' % tpc
    assert np.array_equal(eps.lengths, g[pre + 'lengths'])
    assert np.array_equal([int(s) for s in eps.step_types],
                          g[pre + 'step_types'])
    assert np.array_equal(eps.rewards, g[pre + 'rewards'])
    assert np.array_equal(eps.actions, g[pre + 'actions'])
    assert np.array_equal(eps.observations, g[pre + 'observations'])
    assert np.array_equal(eps.last_observations, g[pre + 'last_observations'])
    assert sampler.total_env_steps == int(np.sum(g[pre + 'lengths']))
