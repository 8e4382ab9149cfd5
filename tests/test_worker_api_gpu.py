"""The stepwise ``Worker`` contract (``sampler/worker.py:46-77``):
``start_episode(); while not step_episode(): pass; collect_episode()`` on
``GpuVecWorker`` / ``GpuFragmentWorker``.

Pinned three ways: against the real ``VecWorker`` / ``FragmentWorker`` goldens
(``tests/golden/sampler.npz``, what the reference's ``LocalSampler`` loop
concatenates from ``worker.rollout()`` calls), against ``rollout()`` of a twin
worker bit for bit, and against ``rollout_samples()`` (the one-launch path the
sampler takes for a single worker) on the synthetic env.
"""
import numpy as np
import pytest
import torch

from garage_amd._dtypes import EpisodeBatch

pytestmark = pytest.mark.gpu


def _counting_setup(golden, worker_class, **worker_args):
    from garage_amd._dtypes import Box, EnvSpec
    from garage_amd.policies import GaussianMLPPolicy
    from garage_amd.sampler import GpuVecSampler
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
        worker_class=worker_class,
        worker_args=dict(n_envs=n, noise_fn=noise_fn, **worker_args))
    return g, sampler, sampler._workers[0]


def _stepwise(worker, num_samples):
    """``LocalSampler.obtain_samples``'s loop (``local_sampler.py:157-166``) with
    ``rollout()`` spelled out as the three Worker calls."""
    batches, done = [], 0
    while done < num_samples:
        worker.start_episode()
        while not worker.step_episode():
            pass
        batch = worker.collect_episode()
        done += len(batch.actions)
        batches.append(batch.to_host())
    return EpisodeBatch.concatenate(*batches)


def test_vec_worker_stepwise_matches_real_vecworker(golden):
    from garage_amd.sampler import GpuVecWorker
    g, sampler, worker = _counting_setup(golden, GpuVecWorker)
    for prefix, num in (('vec_', 30), ('vec2_', 17)):
        sampler._update_workers(None, None)  # what obtain_samples does first
        eps = _stepwise(worker, num)
        assert np.array_equal(eps.lengths, g[prefix + 'lengths'])
        assert np.array_equal([int(s) for s in eps.step_types],
                              g[prefix + 'step_types'])
        assert np.array_equal(eps.rewards, g[prefix + 'rewards'])
        assert np.array_equal(eps.actions, g[prefix + 'actions'])
        assert np.array_equal(eps.last_observations,
                              g[prefix + 'last_observations'])


def _same(a, b):
    assert np.array_equal(a.lengths, b.lengths)
    assert np.array_equal([int(s) for s in a.step_types],
                          [int(s) for s in b.step_types])
    for k in ('observations', 'last_observations', 'actions', 'rewards'):
        assert np.array_equal(getattr(a, k), getattr(b, k)), k
    for k in a.agent_infos:
        assert np.array_equal(a.agent_infos[k], b.agent_infos[k]), k


@pytest.mark.parametrize('hidden', [(16, 16), (256, 256)])
def test_vec_worker_stepwise_equals_rollout_and_rollout_samples(hidden):
    """Synthetic env, ragged episodes, device Philox noise: the three ways of
    driving a worker give the same bits -- stepwise calls, ``rollout()``, and
    ``rollout_samples()`` (whole rollout in ONE launch) on a third twin."""
    from garage_amd.envs import SyntheticVecEnv
    from garage_amd.policies import GaussianMLPPolicy
    from garage_amd.sampler import GpuVecSampler, GpuVecWorker
    n, O, A, P = 48, 17, 6, 24

    def make():
        torch.manual_seed(11)
        env = SyntheticVecEnv(n, O, A, P, min_len=4, seed=5)
        pol = GaussianMLPPolicy(env.spec, hidden_sizes=hidden)
        s = GpuVecSampler(pol, env, max_episode_length=P, n_workers=1,
                          worker_class=GpuVecWorker, seed=2,
                          worker_args=dict(n_envs=n))
        return s, s._workers[0]

    (sa, wa), (sb, wb), (sc, wc) = make(), make(), make()
    num = 3 * n * P // 2
    got = _stepwise(wa, num)
    batches, done = [], 0
    while done < num:
        b = wb.rollout()
        done += len(b.actions)
        batches.append(b.to_host())
    want = EpisodeBatch.concatenate(*batches)
    _same(got, want)
    whole = wc.rollout_samples(num).to_host()
    _same(got, whole)
    # steps taken without collecting in between (the buffer grows), then one
    # collection: the same episodes as collecting after every finishing step
    wa.update_agent(None)
    wb.update_agent(None)
    wa.start_episode()
    finished = 0
    for _ in range(3 * P):
        finished += bool(wa.step_episode())
    late = wa.collect_episode().to_host()
    wb.start_episode()
    parts = []
    for _ in range(3 * P):
        if wb.step_episode():
            parts.append(wb.collect_episode().to_host())
    assert finished == len(parts) > 3
    _same(late, EpisodeBatch.concatenate(*parts))
    with pytest.raises(ValueError):
        wa.collect_episode()  # nothing completed since


@pytest.mark.parametrize('tpc', [1, 2])
def test_fragment_worker_stepwise_matches_real_fragment_worker(golden, tpc):
    from garage_amd.sampler import GpuFragmentWorker
    g, sampler, worker = _counting_setup(golden, GpuFragmentWorker,
                                         timesteps_per_call=tpc)
    batches, done = [], 0
    while done < 20:
        worker.start_episode()
        for _ in range(tpc):
            worker.step_episode()
        batch = worker.collect_episode()
        done += len(batch.actions)
        batches.append(batch.to_host())
    eps = EpisodeBatch.concatenate(*batches)
    pre = 'frag%d_' % tpc
    assert np.array_equal(eps.lengths, g[pre + 'lengths'])
    assert np.array_equal([int(s) for s in eps.step_types],
                          g[pre + 'step_types'])
    assert np.array_equal(eps.rewards, g[pre + 'rewards'])
    assert np.array_equal(eps.actions, g[pre + 'actions'])
    assert np.array_equal(eps.observations, g[pre + 'observations'])
    assert np.array_equal(eps.last_observations, g[pre + 'last_observations'])
