"""Kernel-level parity: each HIP entry point against the oracle (GPU box)."""
import ctypes as C
from collections import OrderedDict

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    from garage_amd.engine import require_gpu
    return require_gpu()


def _params(g, prefix):
    out = OrderedDict()
    for k in g.files:
        if k.startswith(prefix):
            out[k[len(prefix):]] = torch.from_numpy(g[k].copy())
    return out


def _load_flat(mlp, params, prefix):
    """Copy an oracle/reference param dict into a FlatMLP."""
    for key, view in mlp.named_views():
        view.copy_(params[prefix + key].to(view.device).reshape(view.shape))


# ---------------------------------------------------------------------------
def test_gae_scan_padded_matches_reference_functions(dev):
    from garage_amd.engine import gae_scan
    from oracle import returns as orr
    rng = np.random.RandomState(0)
    for (N, P, g, lam) in [(5, 6, 0.95, 0.5), (33, 256, 0.99, 0.97),
                           (7, 300, 0.99, 0.97), (64, 128, 1.0, 1.0),
                           (3, 1, 0.9, 0.3), (10, 1000, 0.999, 0.95)]:
        rew = rng.randn(N, P).astype(np.float32)
        val = rng.randn(N, P).astype(np.float32)
        adv, ret = gae_scan(torch.from_numpy(rew).to(dev),
                            torch.from_numpy(val).to(dev), discount=g,
                            gae_lambda=lam, max_episode_length=P)
        want_adv = orr.compute_advantages(g, lam, P, torch.from_numpy(val),
                                          torch.from_numpy(rew)).numpy()
        want_ret = np.stack([orr.discount_cumsum(r, g)
                             for r in rew.astype(np.float64)])
        scale = max(1.0, np.abs(want_adv).max())
        assert np.allclose(adv.cpu().numpy(), want_adv, atol=2e-5 * scale)
        assert np.allclose(ret.cpu().numpy(), want_ret.astype(np.float32),
                           rtol=1e-6, atol=1e-6)


def test_gae_scan_golden_ragged_with_v0(dev, golden):
    """Real compute_advantages outputs on V(0)-padded ragged batches (Q2)."""
    from garage_amd.engine import gae_scan
    g = golden('advantages')
    for k in range(int(g['n_ragged_cases'])):
        rew, base, lens = (g['r%d_rewards' % k], g['r%d_base' % k],
                           g['r%d_lens' % k])
        d, lam, P, v0 = g['r%d_cfg' % k]
        P = int(P)
        vals = np.concatenate([base[i, :L] for i, L in enumerate(lens)])
        rews = np.concatenate([rew[i, :L] for i, L in enumerate(lens)])
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        adv, ret = gae_scan(torch.from_numpy(rews).to(dev),
                            torch.from_numpy(vals).to(dev), discount=float(d),
                            gae_lambda=float(lam), max_episode_length=P,
                            offsets=torch.from_numpy(off).to(dev),
                            max_len=int(lens.max()), v0=float(v0))
        want = np.concatenate([g['r%d_adv' % k][i, :L]
                               for i, L in enumerate(lens)])
        assert np.allclose(adv.cpu().numpy(), want, atol=1e-5), k


def test_gae_scan_rollout_buffer_tails(dev):
    """mode 0: several episodes per env row, marked by tail lengths."""
    from garage_amd.engine import gae_scan
    from oracle import returns as orr
    rng = np.random.RandomState(3)
    for (n, T, P) in [(9, 64, 16), (130, 256, 256), (4, 520, 100), (17, 36, 7)]:
        rew = rng.randn(n, T).astype(np.float32)
        val = rng.randn(n, T).astype(np.float32)
        tail = np.zeros((n, T), np.uint16)
        eps = []
        for i in range(n):
            t = 0
            while True:
                L = int(rng.randint(1, P + 1))
                if t + L > T:
                    break
                tail[i, t + L - 1] = L
                eps.append((i, t, L))
                t += L
        v0, bonus = 0.37, 0.11
        adv, ret = gae_scan(torch.from_numpy(rew).to(dev),
                            torch.from_numpy(val).to(dev), discount=0.99,
                            gae_lambda=0.95, max_episode_length=P,
                            tail=torch.from_numpy(tail).to(dev), v0=v0,
                            bonus_const=bonus)
        adv, ret = adv.cpu().numpy(), ret.cpu().numpy()
        for (i, t, L) in eps:
            # reference semantics on the padded row of this episode
            r = np.zeros((1, P), np.float32)
            b = np.full((1, P), v0, np.float32)
            r[0, :L] = rew[i, t:t + L]
            b[0, :L] = val[i, t:t + L]
            want = orr.gae_padded_f64(np.float32(0.99),
                                      float(np.float32(0.99 * 0.95)) /
                                      float(np.float32(0.99)), b,
                                      r + np.float32(bonus))[0, :L]
            assert np.allclose(adv[i, t:t + L], want, atol=2e-5)
            want_ret = orr.discount_cumsum(rew[i, t:t + L].astype(np.float64),
                                           0.99)
            assert np.allclose(ret[i, t:t + L], want_ret, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('n,T', [(1, 4), (7, 64), (300, 128), (33, 200),
                                 (4096, 256)])
def test_gae_scan_fast_path_fixed_horizon_equals_general_kernel(dev, n, T):
    """Whole episodes of exactly P steps take the constant-decay kernel; the
    general (segmented, padded-tail aware) kernel must give the same numbers."""
    from garage_amd import _lib
    from garage_amd.engine import gae_scan
    lib = _lib.load()
    g = torch.Generator(device='cpu').manual_seed(T + n)
    r = torch.randn(n, T, generator=g).to(dev)
    v = torch.randn(n, T, generator=g).to(dev)
    out = {}
    for on in (0, 1):
        lib.ga_set_gae_fixed_fast_path(on)
        out[on] = gae_scan(r, v, discount=0.99, gae_lambda=0.97,
                           max_episode_length=T, v0=0.3, bonus_const=0.05)
    lib.ga_set_gae_fixed_fast_path(1)
    for a, b in zip(out[0], out[1]):
        scale = max(1.0, float(a.abs().max()))
        assert torch.allclose(a, b, atol=2e-7 * scale, rtol=0), (n, T)


@pytest.mark.parametrize('P', [40, 256])
def test_gae_scan_ragged_fast_path_equals_general_kernel(dev, P):
    """Packed ragged whole-episode rows (arbitrary, unaligned starts; V(0) != 0 in
    the padded tail; a constant reward bonus) and padded rows shorter than P:
    the constant-decay kernel against the general segmented one."""
    from garage_amd import _lib
    from garage_amd.engine import gae_scan
    lib = _lib.load()
    rng = np.random.RandomState(P)
    lens = rng.randint(1, P + 1, size=777)
    lens[:5] = [1, 2, 3, P, P]
    off = np.concatenate([[0], np.cumsum(lens)])
    S = int(off[-1])
    g = torch.Generator(device='cpu').manual_seed(P)
    r = torch.randn(S, generator=g).to(dev)
    v = torch.randn(S, generator=g).to(dev)
    offsets = torch.from_numpy(off).to(dev)
    r2 = torch.randn(50, P - 8, generator=g).to(dev)   # padded rows, L = P - 8
    v2 = torch.randn(50, P - 8, generator=g).to(dev)
    out = {}
    for on in (0, 1):
        lib.ga_set_gae_fixed_fast_path(on)
        a = gae_scan(r, v, discount=0.99, gae_lambda=0.97, max_episode_length=P,
                     offsets=offsets, max_len=int(lens.max()), v0=0.37,
                     bonus_const=0.05)
        b = gae_scan(r2, v2, discount=0.99, gae_lambda=0.97,
                     max_episode_length=P, v0=-0.6)
        out[on] = [t.clone() for t in a + b]
    lib.ga_set_gae_fixed_fast_path(1)
    for x, y in zip(out[0], out[1]):
        scale = max(1.0, float(x.abs().max()))
        assert torch.allclose(x, y, atol=3e-7 * scale, rtol=0), P


def test_gemm_nt(dev):
    from garage_amd._lib import call, dptr, stream_ptr
    rng = np.random.RandomState(1)
    for (M, N, K) in [(128, 128, 32), (300, 256, 256), (1000, 6, 256),
                      (77, 200, 20), (4096, 256, 17 + 3), (5, 3, 4),
                      (1024, 256, 376), (640, 384, 400), (256, 128, 36)]:
        A = rng.randn(M, K).astype(np.float32)
        B = rng.randn(N, K).astype(np.float32)
        a, b = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
        c = torch.zeros(M, N, device=dev)
        call('ga_gemm_nt_f32', dptr(a), K, dptr(b), K, dptr(c), N, M, N, K,
             stream_ptr())
        want = A.astype(np.float64) @ B.astype(np.float64).T
        assert np.allclose(c.cpu().numpy(), want, atol=1e-4 * np.sqrt(K)), \
            (M, N, K)


@pytest.mark.parametrize('tag', ['tiny', 'c2', 'c3', 'deep'])
def test_mlp_forward_backward_vs_autograd(dev, golden, tag):
    from garage_amd.engine import FlatMLP, pad_rows
    from oracle import networks as nets
    g = golden('networks')
    pol = _params(g, tag + '_pol:')
    hs = [int(v) for v in g[tag + '_hidden']]
    obs = g[tag + '_obs']
    O, A = obs.shape[1], g[tag + '_act'].shape[1]
    mlp = FlatMLP(O, A, hs, dev)
    _load_flat(mlp, pol, '_module.')
    rng = np.random.RandomState(2)
    # a larger batch so the 128/256-row tiles and split-K are exercised
    X = np.concatenate([obs, rng.randn(1500, O).astype(np.float32)])
    M = X.shape[0]
    Xd = pad_rows(X)
    out = mlp.forward(Xd, M)
    with torch.no_grad():
        want = nets.mlp_mean(pol, '_module.', torch.from_numpy(X))
    assert np.allclose(out[:, :A].cpu().numpy(), want.numpy(), atol=2e-6)
    assert np.allclose(out[:37, :A].cpu().numpy(), g[tag + '_mean'], atol=2e-6)
    # backward of L = sum(out * G)
    G = rng.randn(M, A).astype(np.float32)
    params = OrderedDict((k, v.clone().requires_grad_(v.is_floating_point()
                                                      and 'min_std' not in k))
                         for k, v in pol.items())
    loss = (nets.mlp_mean(params, '_module.', torch.from_numpy(X)) *
            torch.from_numpy(G)).sum()
    loss.backward()
    dout = mlp.dout_view(M)
    dout.zero_()
    dout[:, :A] = torch.from_numpy(G).to(dev)
    mlp.backward(Xd, M, dout)
    mlp.reduce_grads()
    for key, view in mlp.named_views(mlp.grads):
        if key == '_init_std':
            continue
        ref = params['_module.' + key].grad.numpy()
        got = view.cpu().numpy().reshape(ref.shape)
        tol = 1e-5 * max(1.0, np.abs(ref).max())
        assert np.allclose(got, ref, atol=tol), key


def test_mlp_forward_gathered_rows(dev):
    from garage_amd.engine import FlatMLP, pad_rows
    rng = np.random.RandomState(5)
    mlp = FlatMLP(17, 6, (32, 32), dev)
    mlp.params.copy_(torch.from_numpy(
        (rng.randn(mlp.n_flat) * 0.2).astype(np.float32)))
    # padding columns of the weights must be zero for the layout contract
    for l in range(3):
        w = mlp.params[mlp.w_off[l]:mlp.b_off[l]].view(mlp.dims[l + 1], -1)
        w[:, mlp.dims[l]:] = 0
    X = pad_rows(rng.randn(500, 17).astype(np.float32))
    idx = torch.from_numpy(rng.permutation(500)[:333].astype(np.int32)).to(dev)
    a = mlp.forward(X, 333, row_idx=idx).clone()
    b = mlp.forward(X[idx.long()].contiguous(), 333).clone()
    assert torch.equal(a, b)


@pytest.mark.parametrize('shape', [(17, 6, (256, 256)), (4, 2, (64, 64)),
                                   (3, 1, (32, 128)), (24, 17, (512,)),
                                   (32, 24, (256, 64)), (8, 1, (1024, 16))])
def test_streaming_layer_kernels_match_mfma_tiles(dev, shape):
    """The HBM-streaming kernels for the narrow layer products (skinny.hip) against
    the MFMA tile kernels on the same inputs: forward, gathered rows, ragged row
    counts, several gradient splits."""
    from garage_amd import _lib
    from garage_amd.engine import FlatMLP, pad_rows
    lib = _lib.load()
    O, A, hs = shape
    rng = np.random.RandomState(11)
    mlp = FlatMLP(O, A, hs, dev)
    mlp.params.copy_(torch.from_numpy(
        (rng.randn(mlp.n_flat) * 0.3).astype(np.float32)))
    for l in range(len(hs) + 1):
        w = mlp.params[mlp.w_off[l]:mlp.b_off[l]].view(mlp.dims[l + 1], -1)
        w[:, mlp.dims[l]:] = 0
    n_rows = 3000
    X = pad_rows(rng.randn(n_rows, O).astype(np.float32))
    for M, gather in ((1, False), (777, True), (2999, True), (3000, False)):
        idx = None
        if gather:
            idx = torch.from_numpy(
                rng.randint(0, n_rows, size=M).astype(np.int32)).to(dev)
        G = rng.randn(M, A).astype(np.float32)
        res = {}
        for on in (0, 1):
            lib.ga_set_skinny_kernels(on)
            out = mlp.forward(X, M, row_idx=idx).clone()
            dout = mlp.dout_view(M)
            dout.fill_(float('nan'))  # padding columns must never be read
            dout[:, :A] = torch.from_numpy(G).to(dev)
            mlp.backward(X, M, dout, row_idx=idx)
            mlp.reduce_grads()
            res[on] = (out[:, :A].clone(), mlp.grads.clone())
        lib.ga_set_skinny_kernels(1)
        assert torch.isfinite(res[1][0]).all() and torch.isfinite(res[1][1]).all()
        # different summation orders of the same fp32 products: a few ulp of the
        # largest term
        scale = max(1.0, float(res[0][0].abs().max()))
        assert torch.allclose(res[0][0], res[1][0], atol=5e-6 * scale,
                              rtol=1e-5), M
        scale = max(1.0, float(res[0][1].abs().max()))
        assert torch.allclose(res[0][1], res[1][1], atol=5e-6 * scale,
                              rtol=1e-5), M


@pytest.mark.parametrize('shape', [(17, 6, (256, 256)), (40, 3, (512, 96, 130)),
                                   (70, 17, (512, 512))])
def test_small_m_gemm_dispatch_is_bit_identical(dev, shape):
    """GEMMs of few row tiles run on 64x64 tiles with 128-deep k-steps
    (``ga_set_small_m_gemm``): every output element accumulates its k in the same
    order as on the 128x128 tiles, so forward, activations and gradients have the
    same bits, for ragged and gathered row counts."""
    from garage_amd import _lib
    from garage_amd.engine import FlatMLP, pad_rows
    import os
    if os.environ.get('GARAGE_AMD_SPLIT_BF16') == '1':
        pytest.skip('compares two EXACT tile shapes bit for bit; with the opt-in '
                    'split-operand k-loops the 128 x 128 side is not an exact kernel')
    lib = _lib.load()
    O, A, hs = shape
    rng = np.random.RandomState(7)
    mlp = FlatMLP(O, A, hs, dev)
    mlp.params.copy_(torch.from_numpy(
        (rng.randn(mlp.n_flat) * 0.1).astype(np.float32)))
    n_rows = 3000
    X = pad_rows(rng.randn(n_rows, O).astype(np.float32))
    for M, gather in ((64, False), (1, False), (333, True), (2999, True)):
        idx = None
        if gather:
            idx = torch.from_numpy(
                rng.randint(0, n_rows, size=M).astype(np.int32)).to(dev)
        G = rng.randn(M, A).astype(np.float32)
        res = {}
        for on in (0, 1):
            lib.ga_set_small_m_gemm(on)
            out = mlp.forward(X, M, row_idx=idx)[:, :A].clone()
            dout = mlp.dout_view(M)
            dout.zero_()
            dout[:, :A] = torch.from_numpy(G).to(dev)
            mlp.backward(X, M, dout, row_idx=idx)
            mlp.reduce_grads()
            res[on] = (out, mlp._acts.clone(), mlp.grads.clone())
        lib.ga_set_small_m_gemm(1)
        for a0, a1 in zip(res[0], res[1]):
            assert torch.equal(a0, a1), M


@pytest.mark.parametrize('shape', [(17, 6, (256, 256)), (4, 2, (64, 64)),
                                   (9, 1, (64, 128)), (17, 8, (128, 256)),
                                   (40, 3, (256,)), (6, 1, (32, 64))])
def test_head_in_gemm_epilogue_matches_separate_head_launch(dev, shape):
    """``ga_mlp_forward_f32`` with the head layer applied in the epilogue of the
    last hidden layer's GEMM (64 / 128 / 256-wide layers, M a multiple of 64)
    against the separate narrow head launch: same hidden activations bit for bit,
    head outputs to summation-order rounding; other M take the separate launch."""
    from garage_amd import _lib
    from garage_amd.engine import FlatMLP, pad_rows
    lib = _lib.load()
    O, A, hs = shape
    rng = np.random.RandomState(5)
    mlp = FlatMLP(O, A, hs, dev)
    mlp.params.copy_(torch.from_numpy(
        (rng.randn(mlp.n_flat) * 0.2).astype(np.float32)))
    n_rows = 4096
    X = pad_rows(rng.randn(n_rows, O).astype(np.float32))
    for M, gather in ((64, False), (1024, True), (4096, False), (4032, True),
                      (1000, True)):
        idx = None
        if gather:
            idx = torch.from_numpy(
                rng.randint(0, n_rows, size=M).astype(np.int32)).to(dev)
        res = {}
        for on in (0, 1):
            lib.ga_set_fused_head_forward(2 * on)
            mlp._workspace(n_rows)
            mlp.out_view(M).fill_(float('nan'))
            out = mlp.forward(X, M, row_idx=idx)[:, :A].clone()
            ldh = (hs[-1] + 3) // 4 * 4
            off = mlp.act_off[len(hs) - 1] * mlp._cap
            res[on] = (out, mlp._acts[off:off + M * ldh].clone())
        lib.ga_set_fused_head_forward(1)
        assert torch.isfinite(res[1][0]).all(), M
        assert torch.equal(res[0][1], res[1][1]), M
        scale = max(1.0, float(res[0][0].abs().max()))
        assert torch.allclose(res[0][0], res[1][0], atol=5e-6 * scale,
                              rtol=1e-5), M
        if M % 64:
            assert torch.equal(res[0][0], res[1][0]), M


@pytest.mark.parametrize('hidden,A', [(64, 2), (128, 1), (256, 6), (512, 17)])
@pytest.mark.parametrize('algo', [0, 1, 2])
def test_head_fused_into_loss_matches_unfused(dev, hidden, A, algo):
    """``ga_head_ppo_gaussian_loss_f32`` / ``ga_head_gaussian_nll_loss_f32`` (the
    head layer computed inside the loss kernel) against head GEMM + loss kernel:
    means / values, loss, gradient seed, log-likelihoods and the log-std slot."""
    from garage_amd._lib import call, dptr, stream_ptr
    from garage_amd.engine import FlatMLP, pad_rows, reduction_workspace
    rng = np.random.RandomState(hidden + A + algo)
    O, n_rows = 9, 1500
    net = FlatMLP(O, A, (32, hidden), dev)
    net.params.copy_(torch.from_numpy(
        (rng.randn(net.n_flat) * 0.2).astype(np.float32)))
    for l in range(3):
        w = net.params[net.w_off[l]:net.b_off[l]].view(net.dims[l + 1], -1)
        w[:, net.dims[l]:] = 0
    net.params[0] = -0.3
    X = pad_rows(rng.randn(n_rows, O).astype(np.float32))
    act = pad_rows(rng.randn(n_rows, A).astype(np.float32))
    adv = torch.from_numpy(rng.randn(n_rows).astype(np.float32)).to(dev)
    ret = torch.from_numpy(rng.randn(n_rows).astype(np.float32)).to(dev)
    ws = reduction_workspace(dev)
    for M, gather in ((1000, True), (37, False), (1, False)):
        idx = None
        if gather:
            idx = torch.from_numpy(
                rng.permutation(n_rows)[:M].astype(np.int32)).to(dev)
        net._workspace(M)
        splits = int(net._splits)
        # old log-likelihoods: the current ones plus noise (ratios around 1)
        mean = net.forward(X, M, row_idx=idx).clone()
        old_ll = torch.empty(n_rows, device=dev)
        rows = idx.long() if gather else torch.arange(M, device=dev)
        d = act[rows, :A] - mean[:, :A]
        s = float(net.params[0])
        ll = (-0.5 * d * d * np.exp(-2 * s) - s - 0.9189385332).sum(1)
        old_ll[rows] = ll + 0.1 * torch.from_numpy(
            rng.randn(M).astype(np.float32)).to(dev)
        res = []
        for fused in (False, True):
            net._slabs.zero_()
            dout = net.dout_view(M)
            dout.zero_()
            loss = torch.zeros(1, device=dev)
            ll_out = torch.zeros(M, device=dev)
            if A == 1 and algo == 0:  # the value function's NLL head
                if fused:
                    H, Wh, bh = net.forward_hidden(X, M, row_idx=idx)
                    v = net.out_view(M)
                    v.zero_()
                    call('ga_head_gaussian_nll_loss_f32', dptr(H), H.stride(0),
                         dptr(Wh), dptr(bh), hidden, dptr(v), v.stride(0),
                         dptr(ret), dptr(idx), dptr(net.params[0:1]), M,
                         dptr(dout), dout.stride(0), dptr(loss),
                         dptr(net._slabs), net.n_flat, splits, dptr(ws),
                         stream_ptr())
                else:
                    v = net.forward(X, M, row_idx=idx)
                    call('ga_gaussian_nll_loss_f32', dptr(v), v.stride(0),
                         dptr(ret), dptr(idx), dptr(net.params[0:1]), M,
                         dptr(dout), dptr(loss), dptr(net._slabs), net.n_flat,
                         splits, dptr(ws), stream_ptr())
                head = v
            else:
                args = (dptr(act), act.stride(0), dptr(old_ll), dptr(adv),
                        dptr(idx), dptr(net.params[0:1]), 1, -1.0, 0, 0.0, M,
                        A, algo, 0.2, 0.01, 1)
                if fused:
                    H, Wh, bh = net.forward_hidden(X, M, row_idx=idx)
                    head = net.out_view(M)
                    head.zero_()
                    call('ga_head_ppo_gaussian_loss_f32', dptr(H), H.stride(0),
                         dptr(Wh), H.stride(0), dptr(bh), hidden, dptr(head),
                         head.stride(0), *args, dptr(dout), dout.stride(0),
                         dptr(ll_out), dptr(loss), dptr(net._slabs),
                         net.n_flat, splits, dptr(ws), stream_ptr())
                else:
                    head = net.forward(X, M, row_idx=idx)
                    call('ga_ppo_gaussian_loss_f32', dptr(head),
                         head.stride(0), *args, dptr(dout), dptr(ll_out),
                         dptr(loss), dptr(net._slabs), net.n_flat, splits,
                         dptr(ws), stream_ptr())
            res.append((head[:, :A].clone(), loss.clone(),
                        dout[:, :A].clone(), ll_out.clone(),
                        net._slabs[0].clone()))
        (h0, l0, d0, ll0, s0), (h1, l1, d1, ll1, s1) = res
        hs = max(1.0, float(h0.abs().max()))
        assert torch.allclose(h0, h1, atol=5e-6 * hs), (M, 'head')
        assert torch.allclose(l0, l1, rtol=2e-5, atol=1e-6), (M, l0, l1)
        ds = max(1e-6, float(d0.abs().max()))
        assert torch.allclose(d0, d1, atol=2e-4 * ds), (M, 'dout')
        assert torch.allclose(ll0, ll1, rtol=1e-5, atol=2e-4), (M, 'll')
        # (a sum of ~M terms of order 1 that cancels to ~1e-2: rounding of the
        # two head evaluations shows up at 1e-3 of the result)
        assert torch.allclose(s0, s1, rtol=5e-3, atol=2e-5), (M, s0, s1)


def _ppo_oracle_loss(pol, obs, act, old_ll, adv, clip, algo='ppo', ent=None):
    from oracle import networks as nets
    dist = nets.gaussian_dist(pol, '_module.', obs)
    ll = dist.log_prob(act)
    if algo == 'vpg':
        obj = ll * adv
    else:
        ratio = (ll - old_ll).exp()
        obj = torch.min(ratio * adv,
                        torch.clamp(ratio, 1 - clip, 1 + clip) * adv)
    if ent is not None:
        coeff, softplus = ent
        e = dist.entropy()
        if softplus:
            e = torch.nn.functional.softplus(e)
        obj = obj + coeff * e
    return -obj.mean(), ll


@pytest.mark.parametrize('case', ['fresh', 'moved', 'vpg', 'ent', 'entsp'])
def test_ppo_gaussian_loss_and_grads(dev, golden, case):
    from garage_amd._lib import call, dptr, stream_ptr
    from garage_amd.engine import FlatMLP, pad_rows, reduction_workspace
    g = golden('networks')
    tag = 'tiny'
    pol = _params(g, tag + '_pol:')
    hs = [int(v) for v in g[tag + '_hidden']]
    rng = np.random.RandomState(7)
    M, O, A = 700, 4, 2
    obs = torch.from_numpy(rng.randn(M, O).astype(np.float32))
    act = torch.from_numpy(rng.randn(M, A).astype(np.float32))
    adv = torch.from_numpy(rng.randn(M).astype(np.float32))
    params = OrderedDict((k, v.clone().requires_grad_('min_std' not in k))
                         for k, v in pol.items())
    from oracle import networks as nets
    with torch.no_grad():
        old_ll = nets.gaussian_dist(pol, '_module.', obs).log_prob(act)
    if case != 'fresh':
        # old policy differs: ratios leave the clip range for many samples
        old_ll = old_ll + torch.from_numpy(
            (rng.randn(M) * 0.3).astype(np.float32))
    algo = 'vpg' if case == 'vpg' else 'ppo'
    ent = {'ent': (0.02, False), 'entsp': (0.02, True)}.get(case)
    loss, ll = _ppo_oracle_loss(params, obs, act, old_ll, adv, 0.2, algo, ent)
    loss.backward()

    mlp = FlatMLP(O, A, hs, dev)
    _load_flat(mlp, pol, '_module.')
    X = pad_rows(obs)
    actd, advd, olld = pad_rows(act), adv.to(dev), old_ll.to(dev)
    mean = mlp.forward(X, M)
    dout = mlp.dout_view(M)
    ll_out = torch.empty(M, device=dev)
    loss_out = torch.zeros(1, device=dev)
    ws = reduction_workspace(dev)
    flags = 0
    if ent is not None:
        flags = 1 | (2 if ent[1] else 0)
    min_ls = float(pol['_module.min_std_param'])
    mlp._workspace(M)
    splits = int(mlp._splits)
    call('ga_ppo_gaussian_loss_f32', dptr(mean), mean.stride(0), dptr(actd),
         actd.stride(0), dptr(olld), dptr(advd), None, dptr(mlp.log_std), 1,
         min_ls, 0, 0.0, M, A, 1 if algo == 'vpg' else 0, 0.2,
         ent[0] if ent else 0.0, flags, dptr(dout), dptr(ll_out),
         dptr(loss_out), dptr(mlp._slabs), mlp.n_flat, splits, dptr(ws),
         stream_ptr())
    used = mlp.backward(X, M, dout)
    mlp.reduce_grads()
    assert np.isclose(loss_out.item(), loss.item(), atol=1e-6, rtol=1e-5)
    assert np.allclose(ll_out.cpu().numpy(), ll.detach().numpy(), atol=1e-5)
    for key, view in mlp.named_views(mlp.grads):
        ref = params['_module.' + key].grad.numpy()
        got = view.cpu().numpy().reshape(ref.shape)
        assert np.allclose(got, ref, atol=2e-6 + 1e-4 * np.abs(ref).max()), key


def test_value_nll_loss_and_grads(dev, golden):
    from garage_amd._lib import call, dptr, stream_ptr
    from garage_amd.engine import FlatMLP, pad_rows, reduction_workspace
    from oracle import networks as nets
    g = golden('networks')
    vf = _params(g, 'tiny_vf:')
    rng = np.random.RandomState(8)
    M, O = 513, 4
    obs = torch.from_numpy(rng.randn(M, O).astype(np.float32))
    ret = torch.from_numpy(rng.randn(M).astype(np.float32))
    params = OrderedDict((k, v.clone().requires_grad_(True))
                         for k, v in vf.items())
    loss = nets.value_loss(params, obs, ret)
    loss.backward()
    mlp = FlatMLP(O, 1, [8, 8], dev)
    _load_flat(mlp, vf, 'module.')
    X = pad_rows(obs)
    v = mlp.forward(X, M)
    assert np.allclose(v[:, 0].cpu().numpy(),
                       nets.value_forward(vf, obs).detach().numpy(), atol=1e-6)
    dout = mlp.dout_view(M)
    dout.zero_()
    loss_out = torch.zeros(1, device=dev)
    ws = reduction_workspace(dev)
    call('ga_gaussian_nll_loss_f32', dptr(v), v.stride(0), dptr(ret.to(dev)),
         None, dptr(mlp.log_std), M, dptr(dout), dptr(loss_out),
         dptr(mlp._slabs), mlp.n_flat, int(mlp._splits), dptr(ws),
         stream_ptr())
    mlp.backward(X, M, dout)
    mlp.reduce_grads()
    assert np.isclose(loss_out.item(), loss.item(), atol=1e-6, rtol=1e-5)
    for key, view in mlp.named_views(mlp.grads):
        ref = params['module.' + key].grad.numpy()
        got = view.cpu().numpy().reshape(ref.shape)
        assert np.allclose(got, ref, atol=2e-6 + 1e-4 * np.abs(ref).max()), key


def test_adam_matches_torch(dev):
    from garage_amd.engine import FlatMLP
    rng = np.random.RandomState(9)
    mlp = FlatMLP(5, 3, (16, ), dev)
    p0 = (rng.randn(mlp.n_flat) * 0.3).astype(np.float32)
    mlp.params.copy_(torch.from_numpy(p0))
    ref = torch.from_numpy(p0.copy()).requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=2.5e-4)
    for step in range(5):
        gnp = rng.randn(mlp.n_flat).astype(np.float32)
        mlp.grads.copy_(torch.from_numpy(gnp))
        mlp.adam_step(2.5e-4)
        ref.grad = torch.from_numpy(gnp.copy())
        opt.step()
        assert np.allclose(mlp.params.cpu().numpy(), ref.detach().numpy(),
                           atol=1e-7, rtol=1e-6), step
    st = opt.state[ref]
    assert np.allclose(mlp.exp_avg.cpu().numpy(), st['exp_avg'].numpy(),
                       atol=1e-8, rtol=1e-6)
    assert np.allclose(mlp.exp_avg_sq.cpu().numpy(), st['exp_avg_sq'].numpy(),
                       atol=1e-10, rtol=1e-6)


def test_center_advantages(dev, golden):
    from garage_amd.engine import center_advantages
    rng = np.random.RandomState(10)
    x = (rng.randn(100003) * 3 + 1).astype(np.float32)
    for center, positive in [(True, False), (True, True), (False, True)]:
        t = torch.from_numpy(x.copy())
        want = t.clone()
        if center:
            want = (want - want.mean()) / (want.var() + 1e-8)
        if positive:
            want = want - want.min()
        got = center_advantages(t.to(dev), center=center, positive=positive)
        assert np.allclose(got.cpu().numpy(), want.numpy(), atol=1e-5,
                           rtol=1e-5)
    one = center_advantages(torch.tensor([1.5], device=dev))
    assert torch.isnan(one).all()  # var of one element: NaN, as in the reference


def test_gaussian_kl(dev):
    from garage_amd._lib import call, dptr, stream_ptr
    from garage_amd.engine import reduction_workspace
    from torch.distributions import Independent, Normal, kl_divergence
    rng = np.random.RandomState(11)
    M, A = 1234, 6
    m0 = torch.from_numpy(rng.randn(M, A).astype(np.float32))
    m1 = m0 + torch.from_numpy((rng.randn(M, A) * 0.1).astype(np.float32))
    s0, s1 = 0.05, -0.12
    want = kl_divergence(
        Independent(Normal(m0, torch.full_like(m0, s0).exp()), 1),
        Independent(Normal(m1, torch.full_like(m1, s1).exp()), 1)).sum().item()
    from garage_amd.engine import pad_rows
    a, b = pad_rows(m0), pad_rows(m1)
    out = torch.zeros(1, dtype=torch.float64, device=dev)
    call('ga_gaussian_kl_f32', dptr(a), dptr(b), a.stride(0), M, A, s0, s1,
         dptr(out), dptr(reduction_workspace(dev)), stream_ptr())
    assert np.isclose(out.item(), want, rtol=1e-5)


@pytest.mark.parametrize('M', [200, 257, 70001, 1048576])
def test_losses_finish_in_one_launch_with_the_same_bits(dev, M):
    """The loss entry points with the last-ticket finish (one launch) against the
    separate finalize launch: loss and log-std gradient slot bit for bit, for one
    block, a few blocks and the full block count, launch after launch (the
    ticket returns to 0), for the Gaussian PPO, categorical PPO and NLL losses."""
    from garage_amd import _lib
    from garage_amd._lib import call, dptr, stream_ptr
    from garage_amd.engine import reduction_workspace
    lib = _lib.load()
    g = torch.Generator(device='cpu').manual_seed(M)
    A, ld = 3, 4
    mean = torch.randn(M, ld, generator=g).to(dev)
    act = torch.randn(M, ld, generator=g).to(dev)
    act_cls = torch.zeros(M, ld)
    act_cls[:, 0] = torch.randint(0, A, (M, ), generator=g).float()
    act_cls = act_cls.to(dev)
    old_ll = (-2.0 + 0.3 * torch.randn(M, generator=g)).to(dev)
    adv = torch.randn(M, generator=g).to(dev)
    ret = torch.randn(M, generator=g).to(dev)
    log_std = torch.full((4, ), -0.2, device=dev)
    ws = reduction_workspace(dev)
    n_splits, stride = 3, 16
    res = {}
    for on in (0, 1, 1, 0, 1):
        lib.ga_set_one_launch_losses(on)
        out = []
        for kind in ('gauss', 'cat', 'nll'):
            loss = torch.full((1, ), float('nan'), device=dev)
            slabs = torch.full((n_splits * stride, ), float('nan'), device=dev)
            d = torch.zeros(M, ld, device=dev)
            if kind == 'gauss':
                call('ga_ppo_gaussian_loss_f32', dptr(mean), ld, dptr(act), ld,
                     dptr(old_ll), dptr(adv), None, dptr(log_std), 1, -3.0, 0,
                     0.0, M, A, 0, 0.2, 0.01, 1, dptr(d), None, dptr(loss),
                     dptr(slabs), stride, n_splits, dptr(ws), stream_ptr())
            elif kind == 'cat':
                call('ga_ppo_categorical_loss_f32', dptr(mean), ld,
                     dptr(act_cls), ld, dptr(old_ll), dptr(adv), None, M, A, 1,
                     0, 0.2, 0.01, 1, dptr(d), None, None, dptr(loss), None,
                     dptr(slabs), stride, n_splits, dptr(ws), stream_ptr())
            else:
                call('ga_gaussian_nll_loss_f32', dptr(mean), ld, dptr(ret),
                     None, dptr(log_std), M, dptr(d), dptr(loss), dptr(slabs),
                     stride, n_splits, dptr(ws), stream_ptr())
            out.append((loss.clone(), slabs[::stride].clone(), d.clone()))
        torch.cuda.synchronize()
        assert ws[-1].item() == 0.0  # the ticket (last slot) is back at 0
        res.setdefault(on, []).append(out)
    lib.ga_set_one_launch_losses(0)
    base = res[0][0]
    for runs in res.values():
        for out in runs:
            for (l0, s0, d0), (l1, s1, d1) in zip(base, out):
                assert torch.isfinite(l1).all() and torch.isfinite(s1).all()
                assert torch.equal(l0, l1) and torch.equal(s0, s1)
                assert torch.equal(d0, d1)
