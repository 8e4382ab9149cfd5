"""The fused rollout step (policy MLP + head in one launch) against the
per-layer path, and -- through the worker -- against everything the sampler
tests already pin (those run on the fused path by default)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rollout(fused, discrete, O, A, hidden, n, P, steps, noise):
    from garage_amd.envs import SyntheticVecEnv
    from garage_amd.policies import CategoricalMLPPolicy, GaussianMLPPolicy
    from garage_amd.sampler import GpuVecSampler, GpuVecWorker
    torch.manual_seed(21)
    env = SyntheticVecEnv(n, O, A, P, min_len=max(1, P // 3), seed=5,
                          discrete=discrete)
    cls = CategoricalMLPPolicy if discrete else GaussianMLPPolicy
    pol = cls(env.spec, hidden_sizes=hidden)
    with torch.no_grad():  # biases away from zero
        pol.net.params.add_(torch.randn_like(pol.net.params) * 0.05)
        for l in range(len(hidden) + 1):
            w = pol.net.params[pol.net.w_off[l]:pol.net.b_off[l]].view(
                pol.net.dims[l + 1], -1)
            w[:, pol.net.dims[l]:] = 0
    dev = pol.device

    def noise_fn(step):
        return noise[step].to(dev)

    sampler = GpuVecSampler(pol, env, max_episode_length=P, n_workers=1,
                            worker_class=GpuVecWorker,
                            worker_args=dict(n_envs=n, noise_fn=noise_fn,
                                             fused_policy_step=fused))
    return sampler.obtain_samples(0, steps, None)


@pytest.mark.parametrize('discrete,O,A,hidden,n', [
    (False, 17, 6, (256, 256), 300),
    (False, 5, 3, (16, 16), 77),
    (False, 4, 2, (64, 64), 64),
    (False, 33, 7, (40, 24, 100), 45),
    (False, 3, 2, (), 33),
    (True, 4, 2, (64, 64), 130),
    (True, 9, 5, (32, ), 31),
])
def test_fused_step_matches_per_layer_path(discrete, O, A, hidden, n):
    P = 9
    noise = torch.rand(40, n, 8) if discrete else torch.randn(40, n, 8)
    a = _rollout(True, discrete, O, A, hidden, n, P, n * P, noise)
    b = _rollout(False, discrete, O, A, hidden, n, P, n * P, noise)
    assert np.array_equal(a.lengths, b.lengths)
    assert np.array_equal(a.observations, b.observations)
    assert np.array_equal([int(s) for s in a.step_types],
                          [int(s) for s in b.step_types])
    key = 'prob' if discrete else 'mean'
    assert np.allclose(a.agent_infos[key], b.agent_infos[key], atol=2e-6)
    if discrete:
        # identical uniforms; a pick can only differ when u sits within an ulp
        # of a CDF boundary
        assert (a.actions != b.actions).mean() < 0.01
    else:
        assert np.allclose(a.actions, b.actions, atol=2e-6)
        assert np.allclose(a.rewards, b.rewards, atol=2e-6)


@pytest.mark.parametrize('normalize', [False, True])
@pytest.mark.parametrize('ragged', [False, True])
def test_native_rollout_loop_equals_python_driven_steps(ragged, normalize):
    """ga_rollout_synth_steps enqueues the same launches as GpuVecWorker._step:
    with the device RNG both give bit-identical batches, twice in a row (the
    second call exercises the partial reset and the odd/even buffer parity)."""
    from garage_amd.envs import NormalizedVecEnv, SyntheticVecEnv
    from garage_amd.policies import GaussianMLPPolicy
    from garage_amd.sampler import GpuVecSampler, GpuVecWorker

    class PythonSteps(GpuVecWorker):

        def _native_steps(self, b, col, n_steps):
            return False

    n, O, A, P = 70, 6, 3, 11
    out = []
    for cls in (GpuVecWorker, PythonSteps):
        torch.manual_seed(4)
        env = SyntheticVecEnv(n, O, A, P, min_len=3 if ragged else None, seed=8)
        if normalize:  # statistics + normalisation fused into the env step
            env = NormalizedVecEnv(env, normalize_obs=True,
                                   normalize_reward=True, scale_reward=0.5,
                                   obs_alpha=0.05, reward_alpha=0.05)
        pol = GaussianMLPPolicy(env.spec, hidden_sizes=(32, 32))
        sampler = GpuVecSampler(pol, env, max_episode_length=P, n_workers=1,
                                seed=3, worker_class=cls,
                                worker_args=dict(n_envs=n))
        out.append([sampler.obtain_samples(0, num, None)
                    for num in (n * P, n * P + 17)])
    for a, b in zip(*out):
        assert np.array_equal(a.lengths, b.lengths)
        assert np.array_equal(a.observations, b.observations)
        assert np.array_equal(a.actions, b.actions)
        assert np.array_equal(a.rewards, b.rewards)
        assert np.array_equal(a.last_observations, b.last_observations)
        assert np.isfinite(a.observations).all() and a.lengths.sum() > 0
        assert np.array_equal([int(s) for s in a.step_types],
                              [int(s) for s in b.step_types])


@pytest.mark.parametrize('O,A,hidden,M', [(17, 6, (256, 256), 1000),
                                          (17, 1, (256, 256), 4100),
                                          (5, 3, (16, 40), 333),
                                          (33, 2, (100, ), 65)])
def test_fused_training_forward_matches_per_layer_gemms(O, A, hidden, M):
    """ga_mlp_forward_fused_f32 (off by default) == ga_mlp_forward_f32:
    outputs AND the stored hidden activations the backward pass consumes."""
    import ctypes as C

    from garage_amd._lib import call, dptr, load, stream_ptr
    from garage_amd.engine import FlatMLP, pad_rows, require_gpu
    dev = require_gpu()
    rng = np.random.RandomState(0)
    net = FlatMLP(O, A, hidden, dev)
    for l in range(len(hidden) + 1):
        net.weight(l).copy_(torch.from_numpy(
            (rng.randn(net.dims[l + 1], net.dims[l]) * 0.2).astype(np.float32)))
        net.bias(l).copy_(torch.from_numpy(
            (rng.randn(net.dims[l + 1]) * 0.2).astype(np.float32)))
    X = pad_rows(rng.randn(2 * M, O).astype(np.float32))
    idx = torch.from_numpy(rng.permutation(2 * M)[:M].astype(np.int32)).to(dev)
    load().ga_set_fused_forward(0)
    want = net.forward(X, M, row_idx=idx).clone()
    want_acts = net._acts.clone()
    net._acts.zero_()
    out = torch.zeros_like(want)
    call('ga_mlp_forward_fused_f32', C.byref(net._desc), dptr(net.params),
         dptr(X), X.stride(0), dptr(idx), M, dptr(net._acts), dptr(out),
         out.stride(0), stream_ptr())
    assert np.allclose(out[:, :A].cpu().numpy(), want[:, :A].cpu().numpy(),
                       atol=1e-5)
    for l, h in enumerate(hidden):
        w = (h + 3) // 4 * 4
        off = net.act_off[l] * net._cap
        a = net._acts[off:off + M * w].view(M, w)[:, :h]
        b = want_acts[off:off + M * w].view(M, w)[:, :h]
        # (layers the streaming kernels take sum k in a different order than the
        # MFMA slot map the fused kernel shares with the tile GEMMs)
        assert torch.allclose(a, b, atol=5e-6), l
