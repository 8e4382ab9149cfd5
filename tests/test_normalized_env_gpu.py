"""NormalizedVecEnv against the real ``garage.envs.normalize`` golden."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_obs_normalisation_matches_real_normalized_env(golden):
    """tests/golden/normalized_env.npz: 1 reset + 3 steps of the real
    NormalizedEnv around the synthetic env (env 0, seed 9)."""
    from garage_amd.envs import NormalizedVecEnv, SyntheticVecEnv
    g = golden('normalized_env')
    env = NormalizedVecEnv(SyntheticVecEnv(1, 3, 2, 5, seed=9),
                           normalize_obs=True, obs_alpha=float(g['alpha']))
    dev = env.device
    act = torch.tensor([[0.25, -0.5, 0., 0.]], device=dev)
    env.reset_all()
    got = [env.obs[0, :3].cpu().numpy().copy()]
    means = [env._obs_mean[0].cpu().numpy().copy()]
    variances = [env._obs_var[0].cpu().numpy().copy()]
    for _ in range(3):
        env.step_all(act)
        env.advance()
        got.append(env.obs[0, :3].cpu().numpy().copy())
        means.append(env._obs_mean[0].cpu().numpy().copy())
        variances.append(env._obs_var[0].cpu().numpy().copy())
    assert np.allclose(np.asarray(got), g['normed'], rtol=1e-6, atol=1e-7)
    assert np.allclose(np.asarray(means), g['means'], rtol=1e-12)
    assert np.allclose(np.asarray(variances), g['variances'], rtol=1e-12)


def test_normalised_rollout_runs_through_the_sampler():
    """Per-env statistics, masked resets, reward scaling: against a numpy
    restatement driven by the raw (un-normalised) twin."""
    from garage_amd.envs import NormalizedVecEnv, SyntheticVecEnv
    from oracle.sampler import NormalizedObs
    n, O, A, P = 6, 4, 2, 5
    raw = SyntheticVecEnv(n, O, A, P, min_len=2, seed=4)
    env = NormalizedVecEnv(SyntheticVecEnv(n, O, A, P, min_len=2, seed=4),
                           normalize_obs=True, scale_reward=0.5)
    dev = env.device
    norms = [NormalizedObs(O) for _ in range(n)]
    # non-zero actions: the reward depends on the env's own observation, which
    # must stay the raw one inside the wrapper (normalized_env.py:134-151 steps
    # the inner env, then normalises what it returned)
    act = torch.zeros(n, 4, device=dev)
    act[:, 0] = 0.7
    act[:, 1] = -0.4
    raw.reset_all()
    env.reset_all()
    want = np.stack([norms[i](raw.obs[i, :O].cpu().numpy()) for i in range(n)])
    assert np.allclose(env.obs[:, :O].cpu().numpy(), want, rtol=1e-6, atol=1e-7)
    for _ in range(7):
        raw.step_all(act)
        env.step_all(act)
        nxt = raw.next_obs[:, :O].cpu().numpy()
        want = np.stack([norms[i](nxt[i]) for i in range(n)])
        assert np.allclose(env.next_obs[:, :O].cpu().numpy(), want, rtol=1e-6,
                           atol=1e-7)
        assert np.allclose(env.reward.cpu().numpy(),
                           0.5 * raw.reward.cpu().numpy())
        done = (raw.step_type >= 2).to(torch.uint8)
        raw.reset_where(done)
        env.reset_where(done)
        nxt = raw.next_obs[:, :O].cpu().numpy()
        d = done.cpu().numpy()
        for i in range(n):
            if d[i]:
                want[i] = norms[i](nxt[i])
        assert np.allclose(env.next_obs[:, :O].cpu().numpy(), want, rtol=1e-6,
                           atol=1e-7)
        raw.advance()
        env.advance()


ACTION_CASES = ('symmetric', 'wide_expected_scale', 'scale_only',
                'upper_unbounded', 'unbounded')


@pytest.mark.parametrize('tag', ACTION_CASES)
def test_action_rescale_and_reward_normalisation_match_real_normalized_env(
        golden, tag):
    """tests/golden/normalized_env_actions.npz: the real ``garage.envs.normalize``
    (``envs/normalized_env.py:90-132,153-164``) around an env that remembers what
    it was stepped with -- finite bounds with two ``expected_action_scale``s,
    ``scale_reward`` with and without ``normalize_reward``, the half-open Box
    the reference's bound test lets through (inf / NaN actions included), and
    an unbounded Box (no rescale).  Here: the same env objects behind
    ``HostVecEnv`` inside ``NormalizedVecEnv``."""
    from garage_amd._dtypes import Box, EnvSpec
    from garage_amd.envs import HostVecEnv, NormalizedVecEnv
    from oracle.envs import ActionEchoEnv
    g = golden('normalized_env_actions')
    P = int(g['P'])
    cfg = g[tag + '_cfg']
    low, high = cfg[0:2].astype(np.float32), cfg[2:4].astype(np.float32)
    scale, norm_r, scale_r = float(cfg[4]), bool(cfg[5]), float(cfg[6])
    spec = EnvSpec(Box(-np.inf, np.inf, (3, )), Box(low, high),
                   max_episode_length=P)
    members = [ActionEchoEnv(1, 2, P), ActionEchoEnv(1, 2, P)]
    env = NormalizedVecEnv(HostVecEnv(members, spec=spec),
                           scale_reward=scale_r, normalize_reward=norm_r,
                           expected_action_scale=scale)
    dev = env.device
    env.reset_all()
    rewards, means, variances = [], [], []
    for t, a in enumerate(g['actions']):
        if t == P:
            env.reset_all()
        act = torch.zeros(2, 4, device=dev)
        act[:, :2] = torch.from_numpy(a).to(dev)
        env.step_all(act)
        env.advance()
        rewards.append(env.reward.cpu().numpy().copy())
        means.append(env._reward_mean.cpu().numpy().copy())
        variances.append(env._reward_var.cpu().numpy().copy())
    for m in members:  # what the wrapped env was stepped with: bit for bit
        got = np.asarray(m.received)
        assert np.array_equal(got, g[tag + '_received'], equal_nan=True)
    rewards = np.asarray(rewards)
    for i in (0, 1):
        assert np.allclose(rewards[:, i], g[tag + '_rewards'], rtol=2e-6,
                           atol=1e-7, equal_nan=True)
    if norm_r:
        assert np.allclose(np.asarray(means)[:, 0], g[tag + '_reward_mean'],
                           rtol=1e-6, atol=1e-9)
        assert np.allclose(np.asarray(variances)[:, 0], g[tag + '_reward_var'],
                           rtol=1e-6, atol=1e-9)


def test_native_rollout_rescales_actions_like_the_stepwise_path():
    """``ga_rollout_synth_steps`` with a bounded action space (rescale launch
    between the policy step and the env step) against the same rollout driven
    step by step from Python: same bits; and the batch keeps the policy's own
    actions, not the rescaled ones (``normalized_env.py:109``)."""
    from garage_amd.envs import NormalizedVecEnv, SyntheticVecEnv
    from garage_amd.policies import GaussianMLPPolicy
    from garage_amd.sampler import GpuVecSampler, GpuVecWorker
    n, O, A, P = 64, 5, 3, 12
    out = []
    for native in (True, False):
        torch.manual_seed(2)
        env = NormalizedVecEnv(
            SyntheticVecEnv(n, O, A, P, min_len=3, seed=6,
                            action_bounds=(-0.3, 0.2)),
            normalize_obs=True, normalize_reward=True,
            expected_action_scale=1.5)
        pol = GaussianMLPPolicy(env.spec, hidden_sizes=(32, 32))
        sampler = GpuVecSampler(pol, env, max_episode_length=P, n_workers=1,
                                worker_class=GpuVecWorker, seed=4,
                                worker_args=dict(n_envs=n))
        if not native:  # same kernels, driven one step at a time from Python
            for w in sampler._workers:
                w._native_steps = lambda *args: False
        eps = sampler.obtain_samples(0, n * P, None)
        out.append((eps.obs_dev.clone(), eps.actions_dev.clone(),
                    eps.rewards_dev.clone(), np.asarray(eps.lengths)))
    assert np.array_equal(out[0][3], out[1][3])
    for a, b in zip(out[0][:3], out[1][:3]):
        assert torch.equal(a, b)
    acts = out[0][1][:, :A]
    assert float(acts.abs().max()) > 0.5  # unclipped policy actions are stored


def test_does_not_modify_action_and_pickles():
    """``tests/garage/envs/test_normalized_env.py:13-32``: stepping leaves the
    caller's action array as it was (the rescaled / clipped copy goes to the wrapped
    env), and a pickle round trip keeps the wrapper's settings and statistics."""
    import pickle

    from garage_amd._dtypes import Box, EnvSpec
    from garage_amd.envs import HostVecEnv, NormalizedVecEnv, SyntheticVecEnv
    from oracle.envs import ActionEchoEnv
    P = 6
    spec = EnvSpec(Box(-np.inf, np.inf, (3, )),
                   Box(np.asarray([-1., -2.], np.float32),
                       np.asarray([1., 2.], np.float32)),
                   max_episode_length=P)
    members = [ActionEchoEnv(1, 2, P), ActionEchoEnv(1, 2, P)]
    env = NormalizedVecEnv(HostVecEnv(members, spec=spec), scale_reward=10.)
    env.reset_all()
    act = torch.zeros(2, 4, device=env.device)
    act[:, :2] = torch.tensor([[3.0, 5.0], [-4.0, 0.5]], device=env.device)
    before = act.clone()
    env.step_all(act)
    env.advance()
    assert torch.equal(act, before)
    # (the wrapped env did get the clipped values)
    assert np.allclose(members[0].received[-1], [1.0, 2.0])
    assert np.allclose(members[1].received[-1], [-1.0, 1.0])
    # device env: pickle round trip keeps the options and the running statistics
    dev_env = NormalizedVecEnv(SyntheticVecEnv(8, 5, 2, 10, seed=3),
                               scale_reward=10., normalize_obs=True,
                               normalize_reward=True)
    dev_env.reset_all()
    for _ in range(3):
        dev_env.step_all(torch.zeros(8, 4, device=dev_env.device))
        dev_env.advance()
    twin = pickle.loads(pickle.dumps(dev_env))
    assert twin._scale_reward == dev_env._scale_reward
    assert torch.equal(twin._obs_mean.cpu(), dev_env._obs_mean.cpu())
    assert torch.equal(twin._reward_var.cpu(), dev_env._reward_var.cpu())
    twin.step_all(torch.zeros(8, 4, device=twin.device))
    twin.advance()
