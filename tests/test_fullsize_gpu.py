"""Parity at BASELINE.json's full sizes through size-independent properties.

The oracle finishes in seconds only at small sizes, so at C3 (4096 envs x T 256,
1 048 576 steps) and at the ragged C5 row counts the HIP path is checked through
properties that hold at any size: the defining recurrences of the returns / GAE
scan, linearity, bookkeeping invariants of the packed batch, spot checks of
the Philox observations against the oracle's generator, permutation-ness of the
device shuffle, exact zeros the algebra demands (KL(old || old), LossBefore +
mean(adv)), gradient additivity over a partition of the batch, and bitwise
run-to-run reproducibility of a whole iteration.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N_ENVS, T, OBS, ACT = 4096, 256, 17, 6
GAMMA, LAM = 0.99, 0.97


@pytest.fixture(scope='module')
def dev():
    from garage_amd.engine import require_gpu
    return require_gpu()


def test_gae_scan_recurrences_full_size(dev):
    """n = 4096 rows of T = 256: A_t = delta_t + gamma*lambda*A_{t+1},
    G_t = r_t + gamma*G_{t+1}, and linearity of both outputs."""
    from garage_amd.engine import gae_scan
    g = torch.Generator(device='cpu').manual_seed(0)
    r = torch.randn(N_ENVS, T, generator=g).to(dev)
    v = torch.randn(N_ENVS, T, generator=g).to(dev)
    adv, ret = gae_scan(r, v, discount=GAMMA, gae_lambda=LAM,
                        max_episode_length=T)
    r64, v64 = r.double(), v.double()
    g32 = float(np.float32(GAMMA))
    gl32 = float(np.float32(np.float32(GAMMA) * np.float32(LAM)))
    v_next = torch.cat([v64[:, 1:], torch.zeros(N_ENVS, 1, device=dev,
                                                dtype=torch.float64)], 1)
    delta = r64 + g32 * v_next - v64
    a = adv.double()
    a_next = torch.cat([a[:, 1:], torch.zeros_like(a[:, :1])], 1)
    res = (a - gl32 * a_next - delta).abs().max().item()
    assert res < 2e-5, res
    G = ret.double()
    G_next = torch.cat([G[:, 1:], torch.zeros_like(G[:, :1])], 1)
    res = (G - GAMMA * G_next - r64).abs().max().item()
    assert res < 2e-5, res
    # linearity: scan(2 r1 - 3 r2, 2 v1 - 3 v2) = 2 scan(r1, v1) - 3 scan(r2, v2)
    r2 = torch.randn(N_ENVS, T, generator=g).to(dev)
    v2 = torch.randn(N_ENVS, T, generator=g).to(dev)
    adv2, ret2 = gae_scan(r2, v2, discount=GAMMA, gae_lambda=LAM,
                          max_episode_length=T)
    adv3, ret3 = gae_scan(2 * r - 3 * r2, 2 * v - 3 * v2, discount=GAMMA,
                          gae_lambda=LAM, max_episode_length=T)
    scale = float(adv3.abs().max())
    assert torch.allclose(adv3, 2 * adv - 3 * adv2, atol=1e-5 * scale)
    scale = float(ret3.abs().max())
    assert torch.allclose(ret3, 2 * ret - 3 * ret2, atol=1e-5 * scale)


def test_gae_scan_ragged_recurrence_c5_rows(dev):
    """Packed ragged rows (8192 episodes, L ~ U{32..256}, P = 256, V(0) = 0):
    the recurrences hold inside every episode and restart at its end."""
    from garage_amd.engine import gae_scan
    rng = np.random.RandomState(0)
    lens = rng.randint(32, 257, size=8192)
    off = np.concatenate([[0], np.cumsum(lens)])
    S = int(off[-1])
    g = torch.Generator(device='cpu').manual_seed(1)
    r = torch.randn(S, generator=g).to(dev)
    v = torch.randn(S, generator=g).to(dev)
    offsets = torch.from_numpy(off).to(dev)
    adv, ret = gae_scan(r, v, discount=GAMMA, gae_lambda=LAM,
                        max_episode_length=256, offsets=offsets, max_len=256,
                        v0=0.0)
    last = torch.zeros(S, dtype=torch.bool, device=dev)
    last[offsets[1:] - 1] = True
    g32 = float(np.float32(GAMMA))
    gl32 = float(np.float32(np.float32(GAMMA) * np.float32(LAM)))

    def nxt(x):
        y = torch.cat([x[1:], torch.zeros_like(x[:1])])
        return torch.where(last, torch.zeros_like(y), y)

    r64, v64, a, G = r.double(), v.double(), adv.double(), ret.double()
    delta = r64 + g32 * nxt(v64) - v64
    assert (a - gl32 * nxt(a) - delta).abs().max().item() < 2e-5
    assert (G - GAMMA * nxt(G) - r64).abs().max().item() < 2e-5


@pytest.fixture(scope='module')
def engine(dev):
    import bench
    algo, sampler, pol, S = bench.build_engine(bench.CONFIGS['c3'], None, seed=1)
    return bench, algo, sampler, pol, S


def test_rollout_bookkeeping_and_philox_spot_checks_c3(dev, engine):
    """One C3 rollout (4096 envs x 256 steps): counts, step types, packing is
    exactly the env-major buffer, sampled observations equal the oracle's
    Philox generator, and a second sampler with the same seed is bitwise equal."""
    bench, algo, sampler, pol, S = engine
    from oracle import envs as oenvs
    eps = sampler.obtain_samples(0, S, None)
    lens = np.asarray(eps.lengths)
    assert lens.shape == (N_ENVS, ) and (lens == T).all()
    assert eps.n_samples == S == N_ENVS * T
    assert sampler.total_env_steps == S
    st = eps.step_types_dev.view(N_ENVS, T)
    assert bool((st[:, 0] == 0).all()) and bool((st[:, -1] == 3).all())
    assert bool((st[:, 1:-1] == 1).all())
    # every episode finishes at the same step, so batch order = env order and
    # row (i, t) is env i at step t of its episode 0
    obs = eps.obs_dev.view(N_ENVS, T, -1)
    rng = np.random.RandomState(3)
    for _ in range(64):
        i, t = int(rng.randint(N_ENVS)), int(rng.randint(T))
        want = oenvs.synthetic_values(1, i, 0, t, oenvs.STREAM_OBS, OBS)
        got = obs[i, t, :OBS].cpu().numpy()
        assert np.array_equal(got, want), (i, t)
    assert torch.isfinite(eps.rewards_dev).all()
    assert float(eps.rewards_dev.std()) > 0.5  # unit-variance noise + shaping
    # idempotence: same seed, fresh objects -> the same bits
    algo2, sampler2, pol2, _ = bench.build_engine(bench.CONFIGS['c3'], None,
                                                  seed=1)
    eps2 = sampler2.obtain_samples(0, S, None)
    assert torch.equal(eps.obs_dev, eps2.obs_dev)
    assert torch.equal(eps.actions_dev, eps2.actions_dev)
    assert torch.equal(eps.rewards_dev, eps2.rewards_dev)


def test_rollout_ragged_invariants_8192_envs(dev):
    """8192 envs with ragged episode lengths (C5's rollout shape, a narrow MLP
    to keep it short): packed counts, step types at episode ends, lengths from
    the oracle's length generator, (completion step, env) order."""
    import bench
    from oracle import envs as oenvs
    cfg = dict(bench.CONFIGS['c5'], obs_dim=24, act_dim=5, hidden=(64, 64))
    algo, sampler, pol, S = bench.build_engine(cfg, None, seed=2)
    eps = sampler.obtain_samples(0, S, None)
    lens = np.asarray(eps.lengths)
    assert lens.min() >= 32 and lens.max() <= 256
    assert int(lens.sum()) == eps.n_samples >= S
    off = np.concatenate([[0], np.cumsum(lens)])
    st = eps.step_types_dev.cpu().numpy()
    assert (st[off[:-1]] == 0).all()                      # FIRST
    ends = st[off[1:] - 1]
    assert ((ends == 3) == (lens == 256)).all()           # TIMEOUT iff L == P
    assert ((ends == 2) == (lens < 256)).all()            # TERMINAL otherwise
    interior = np.ones(len(st), bool)
    interior[off[:-1]] = False
    interior[off[1:] - 1] = False
    assert (st[interior] == 1).all()
    # the first 8192 completed episodes are episode 0 of some env; their lengths
    # must be the generator's, and completion steps must not decrease
    want0 = np.asarray([oenvs.synthetic_length(2, i, 0, 32, 256)
                        for i in range(8192)])
    first = lens[:200]
    assert np.all(np.diff(first) >= 0)   # ordered by completion step (= length)
    assert sorted(want0)[:200] == sorted(first.tolist())


def test_device_permutation_is_a_permutation_at_c3_size(dev, engine):
    bench, algo, sampler, pol, S = engine
    perms = list(algo._policy_optimizer.epoch_permutations(S))
    assert len(perms) == bench.HYPER['epochs']
    ar = torch.arange(S, device=dev, dtype=torch.int32)
    for p in perms[:3]:
        assert p.dtype == torch.int32 and p.numel() == S
        assert torch.equal(torch.sort(p).values, ar)
    assert not torch.equal(perms[0], perms[1])
    assert float((perms[0] == ar).float().mean()) < 1e-3


def _snapshot(algo):
    out = []
    for m in (algo.policy, algo._value_function):
        n = m.net
        out.append((n.params.clone(), n.exp_avg.clone(), n.exp_avg_sq.clone(),
                    n.adam_steps))
    out.append((algo._policy_optimizer._draws, algo._vf_optimizer._draws,
                algo._old_policy.params.clone()))
    return out


def _restore(algo, snap):
    for m, (p, m1, m2, steps) in zip((algo.policy, algo._value_function),
                                     snap[:2]):
        n = m.net
        n.params.copy_(p)
        n.exp_avg.copy_(m1)
        n.exp_avg_sq.copy_(m2)
        n.adam_steps = steps
    algo._policy_optimizer._draws, algo._vf_optimizer._draws, old = snap[2]
    algo._old_policy.params.copy_(old)


def test_full_iteration_exact_zeros_and_bitwise_reproducibility(dev, engine):
    """A whole C3 PPO iteration (640 optimizer steps on 2 streams): KLBefore is
    exactly the padded-cell KL of a policy with itself (0), LossBefore equals
    -mean(adv) (ratio == 1), centred advantages have zero mean and variance
    1/var(raw), and a second run from the same state gives the same bits."""
    bench, algo, sampler, pol, S = engine
    eps = sampler.obtain_samples(1, S, None)
    snap = _snapshot(algo)
    algo._train_once(1, eps)
    tab1 = dict(algo.last_tabular)
    adv = algo.last_tensors['advantages']
    assert tab1['policy/KLBefore'] == 0.0
    assert abs(tab1['policy/LossBefore'] + float(adv.double().mean())) < 1e-6
    assert abs(float(adv.double().mean())) < 1e-6
    assert np.isfinite(list(tab1.values())).all()
    assert 0.0 < tab1['policy/KL'] < 0.1
    assert tab1['vf/LossAfter'] < tab1['vf/LossBefore']
    p1 = pol.net.params.clone()
    v1 = algo._value_function.net.params.clone()
    _restore(algo, snap)
    algo._train_once(1, eps)
    assert torch.equal(pol.net.params, p1)
    assert torch.equal(algo._value_function.net.params, v1)
    assert algo.last_tabular == tab1
    # the serial schedule (one stream) is the same arithmetic
    _restore(algo, snap)
    algo.overlap_updates = False
    algo._train_once(1, eps)
    algo.overlap_updates = True
    assert torch.equal(pol.net.params, p1)
    assert torch.equal(algo._value_function.net.params, v1)


def test_gradient_additivity_over_minibatches_c3(dev, engine):
    """The policy-loss gradient of the full 1 048 576-row batch equals the mean
    of the gradients of the 32 gathered minibatches that partition it."""
    bench, algo, sampler, pol, S = engine
    eps = sampler.obtain_samples(2, S, None)
    batch = algo._to_device_batch(eps)
    net = pol.net
    net._workspace(S)
    g = torch.Generator(device='cpu').manual_seed(4)
    adv = torch.randn(S, generator=g).to(dev)
    old_ll = torch.empty(S, device=dev)
    algo._policy_loss_pass(batch, adv, None, S, None, ll_out=old_ll)
    old_ll += 0.05 * torch.randn(S, generator=g).to(dev)  # ratios around 1

    def grad(M, idx):
        _, _, dout = algo._policy_loss_pass(batch, adv, old_ll, M, idx,
                                            want_grad=True)
        net.backward(batch.obs_dev, M, dout, row_idx=idx)
        net.reduce_grads()
        return net.grads.double().clone()

    full = grad(S, None)
    perm = next(iter(algo._policy_optimizer.epoch_permutations(S)))
    mb = S // 32
    acc = torch.zeros_like(full)
    for k in range(32):
        acc += grad(mb, perm[k * mb:(k + 1) * mb].contiguous())
    acc /= 32
    scale = float(full.abs().max())
    assert scale > 0
    assert float((acc - full).abs().max()) < 2e-5 * scale


def test_scan_recurrences_hold_at_c4_footprint():
    """The whole-episode scan at the C4 footprint (32768 rows x 256 steps, what
    ``bench.py`` reports as ``roofline_gae_scan_c4``): the defining recurrences on
    every cell."""
    from garage_amd.engine import gae_scan
    dev = torch.device('cuda', 0)
    g = torch.Generator(device='cpu').manual_seed(3)
    n, T = 32768, 256
    r = torch.randn(n, T, generator=g).to(dev)
    v = torch.randn(n, T, generator=g).to(dev)
    adv, ret = gae_scan(r, v, discount=0.99, gae_lambda=0.97,
                        max_episode_length=T)
    # G_t = r_t + gamma G_{t+1};  A_t = delta_t + gamma lambda A_{t+1}
    rd, vd, ad, gd = r.double(), v.double(), adv.double(), ret.double()
    res_g = gd[:, :-1] - (rd[:, :-1] + 0.99 * gd[:, 1:])
    vnext = torch.cat([vd[:, 1:], torch.zeros(n, 1, device=dev,
                                              dtype=torch.float64)], 1)
    g32 = float(torch.tensor(0.99, dtype=torch.float32))
    c32 = float(torch.tensor(0.99 * 0.97, dtype=torch.float32))
    delta = rd + g32 * vnext - vd
    res_a = ad[:, :-1] - (delta[:, :-1] + c32 * ad[:, 1:])
    assert float(res_g.abs().max()) < 2e-5 and float(res_a.abs().max()) < 2e-5
    assert float((gd[:, -1] - rd[:, -1]).abs().max()) < 1e-6
