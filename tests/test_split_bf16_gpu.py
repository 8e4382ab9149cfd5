"""The opt-in split-operand k-loops (``ga_set_split_bf16`` / GARAGE_AMD_SPLIT_BF16=1,
include/garage_amd.h): the fused forward + loss, data-gradient, weight-gradient and
evaluation-forward kernels of 256-unit networks on ``v_mfma_f32_32x32x16_bf16`` with
every fp32 operand as three bf16 terms.  The default stays exact fp32; these tests hold
the experiment to the SAME parity bars as the exact kernels (oracle iteration at 1e-6
with linear Adam, bitwise two-stream schedules) and to the fp64 error of the exact
kernels."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture()
def split_mode():
    from garage_amd import _lib
    lib = _lib.load()
    lib.ga_set_split_bf16(1)
    try:
        yield lib
    finally:
        lib.ga_set_split_bf16(0)


def test_split_results_are_close_to_exact_but_not_the_same_bits():
    """One iteration (2 epochs x several minibatches) at the C3 network shape: the
    split kernels are actually taken (the bits differ) and the scalars agree to
    rounding."""
    from garage_amd import _lib
    import test_fused_train_gpu as T
    lib = _lib.load()
    spec, batch = T._problem('c3_shape')
    res = []
    try:
        for on in (0, 1):
            lib.ga_set_split_bf16(on)
            # Adam with beta1 = beta2 = 0, eps = 1: the update is linear in small
            # gradients, so the parameters expose the gradients of every step
            algo, pol, vf = T._algo('c3_shape', spec,
                                    (torch.optim.Adam, dict(lr=1e-2, betas=(0.0, 0.0),
                                                            eps=1.0)), epochs=2)
            np.random.seed(11)
            algo._train_once(0, batch)
            res.append((pol.net.params.clone(), vf.net.params.clone(),
                        dict(algo.last_tabular)))
    finally:
        lib.ga_set_split_bf16(0)
    for i in (0, 1):
        assert not torch.equal(res[0][i], res[1][i])
        assert float((res[0][i] - res[1][i]).abs().max()) < 1e-6
    for k, v in res[0][2].items():
        assert np.isclose(res[1][2][k], v, rtol=2e-6, atol=2e-7), (k, v, res[1][2][k])


def test_split_c3_iteration_matches_the_oracle_at_the_exact_kernels_bar(split_mode):
    """``tests/test_configs_gpu.py``'s linear-Adam C3 iteration against the oracle, at
    its unchanged 1e-6 bound, with the split kernels."""
    import test_configs_gpu as TC
    TC._oracle_iterations('c3', 16, 2, 16 * 256 // 4, 1, 1e-6, linear_adam=True)


def test_split_gradient_error_against_fp64_is_the_exact_kernels(split_mode):
    """Gradients of one optimizer step over 4096 rows against the oracle's loss in
    float64 (autograd).  Weight matrices: the split kernels' rms error is within 1.5x
    the exact kernels' (+ 1e-8 of the tensor's largest gradient); every tensor: no
    element further than 2e-6 of the tensor's largest gradient from fp64 (the exact
    kernels: 1e-6).  (Bias gradients are 256-element column sums: theirs scatter between
    0.6x and 4x the exact kernels' error, profiles/r03_split_error_histogram.json.)"""
    import bench
    import test_configs_gpu as TC
    from oracle import networks as nets
    from oracle.ppo import OraclePPO
    lib = split_mode
    cfg = bench.CONFIGS['c3']
    n, T = 16, cfg['T']
    S = n * T
    got, ref = {}, None
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
                    full = [k for k in params
                            if k.endswith('init_std' if key == '_init_std' else key)]
                    grad = params[full[0]].grad
                    if grad is not None:
                        ref[name + '/' + key] = grad.clone()
    checked = 0
    for key, r in ref.items():
        scale = float(r.abs().max())
        if scale == 0.0 or r.numel() < 64:
            continue
        e = {m: float(((got[m][key] - r) ** 2).mean().sqrt()) / scale
             for m in ('exact', 'split')}
        worst = {m: float((got[m][key] - r).abs().max()) / scale
                 for m in ('exact', 'split')}
        if r.numel() >= 1024:
            assert e['split'] <= 1.5 * e['exact'] + 1e-8, (key, e)
        assert worst['exact'] <= 1e-6 and worst['split'] <= 2e-6, (key, worst)
        checked += 1
    assert checked >= 8


def test_split_trpo_two_stream_schedule_equals_serial_bitwise(split_mode):
    """The value function's split kernels on the side stream next to the policy step's
    exact kernels: same bits as one after the other.  (Before the library was built
    without packed fp32 this failed: ``v_pk_fma_f32 ... op_sel:[0,1,0]`` in the policy's
    first-layer weight-gradient kernel misread an operand whenever a bf16 MFMA ran on
    the same SIMD -- tools/mfma_valu_hazard.hip.)"""
    import bench
    cfg = bench.CONFIGS['c3']
    algo, sampler, pol, S = bench.build_engine(cfg, None, seed=2, algo_name='trpo')
    vf = algo._value_function
    eps = sampler.obtain_samples(0, S, None)
    s0 = (pol.net.params.clone(), vf.net.params.clone(), vf.net.exp_avg.clone(),
          vf.net.exp_avg_sq.clone(), vf.net.adam_steps, algo._vf_optimizer._draws,
          algo._old_policy.params.clone())
    results = []
    for overlap in (True, False, True):
        pol.net.params.copy_(s0[0])
        vf.net.params.copy_(s0[1])
        vf.net.exp_avg.copy_(s0[2])
        vf.net.exp_avg_sq.copy_(s0[3])
        vf.net.adam_steps = s0[4]
        algo._vf_optimizer._draws = s0[5]
        algo._old_policy.params.copy_(s0[6])
        algo.overlap_updates = overlap
        algo._train_once(0, eps)
        torch.cuda.synchronize()
        results.append((pol.net.params.clone(), vf.net.params.clone()))
    for got in results[1:]:
        assert torch.equal(got[0], results[0][0])
        assert torch.equal(got[1], results[0][1])


def test_split_evaluation_forward_matches_the_exact_one(split_mode):
    from garage_amd.engine import FlatMLP
    lib = split_mode
    dev = torch.device('cuda')
    torch.manual_seed(5)
    net = FlatMLP(17, 6, (256, 256), dev)
    net.params.copy_(0.05 * torch.randn(net.params.numel(), device=dev))
    M = 70000  # ragged last tile, above the outputs-only threshold
    X = torch.randn(M, 20, device=dev)
    X[:, 17:] = 0
    lib.ga_set_split_bf16(0)
    want = net.forward(X, M, keep_acts=False)[:, :6].clone()
    lib.ga_set_split_bf16(1)
    got = net.forward(X, M, keep_acts=False)[:, :6].clone()
    assert not torch.equal(got, want)
    assert float((got - want).abs().max()) < 2e-6 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize('shape', [(376, 17, (512, 512, 512), 5000),
                                   (40, 3, (128, 384), 8193)])
def test_split_wide_layers_forward_and_backward_match_the_exact_gemms(split_mode, shape):
    """Layers of 128 units and more take the per-layer GEMMs; with the experiment on
    their forward and data-gradient launches run ``gemm_kc_split_kernel`` (ragged row
    tiles, K = 376 -- not a multiple of 32 --, N = 384 -- three column tiles) and the
    128-aligned weight gradients ``gemm_nt_split_kernel``.  Outputs and gradients against
    the exact kernels; a float64 torch reference bounds both."""
    from garage_amd.engine import FlatMLP
    lib = split_mode
    in_dim, out_dim, hidden, M = shape
    dev = torch.device('cuda')
    torch.manual_seed(7)
    net = FlatMLP(in_dim, out_dim, hidden, dev)
    for key, view in net.named_views():
        view.copy_(torch.randn(view.shape, device=dev) / (view.shape[-1] ** 0.5))
    ldx = (in_dim + 3) // 4 * 4
    X = torch.zeros(M, ldx, device=dev)
    X[:, :in_dim] = torch.randn(M, in_dim, device=dev)
    got = {}
    for mode in (0, 1):
        lib.ga_set_split_bf16(mode)
        out = net.forward(X, M, keep_acts=True)[:, :out_dim].clone()
        d = net.dout_view(M)
        d.zero_()
        torch.manual_seed(9)
        d[:, :out_dim] = torch.randn(M, out_dim, device=dev) / M
        net.backward(X, M, d)
        net.reduce_grads(scale=1.0)
        got[mode] = (out, {k: v.clone() for k, v in net.named_views(net.grads)})
    # float64 reference
    h = X[:, :in_dim].double()
    params = {k: v.double().requires_grad_(True) for k, v in net.named_views()}
    names = [k for k in params if k.endswith('weight')]
    for i, w in enumerate(names):
        b = w[:-len('weight')] + 'bias'
        h = h @ params[w].t() + params[b]
        if i + 1 < len(names):
            h = torch.tanh(h)
    torch.manual_seed(9)
    dref = (torch.randn(M, out_dim, device=dev) / M).double()
    (h * dref).sum().backward()
    assert not torch.equal(got[0][0], got[1][0])
    oscale = float(h.abs().max())
    for mode in (0, 1):
        assert float((got[mode][0].double() - h).abs().max()) < 3e-6 * oscale
    for k, p in params.items():
        if p.grad is None:
            continue
        scale = float(p.grad.abs().max())
        e = [float((got[m][1][k].double() - p.grad).abs().max()) / scale for m in (0, 1)]
        assert e[0] < 5e-6 and e[1] < 5e-6, (k, e)
        # (weight matrices: the exact kernels' error; bias gradients -- column sums of a
        # split-operand product -- up to 3x)
        assert e[1] < (3.0 if k.endswith('bias') else 2.0) * e[0] + 5e-7, (k, e)


def test_backward_pass_keeps_its_bits_next_to_bf16_mfmas():
    """The hazard of ``tools/mfma_valu_hazard.hip`` seen from the library: a forward +
    backward pass of a C3-shaped network (first-layer weight gradient on the streaming
    kernel hipcc used to vectorize into ``v_pk_fma_f32 ... op_sel:[0,1,0]``) while a
    second stream runs a register-only bf16 MFMA loop must give the bits it gives alone.
    Built with the SLP vectorizer this changed ~60 000 of 655 360 slab elements."""
    import ctypes as C
    from garage_amd import _lib
    from garage_amd.engine import FlatMLP
    lib = _lib.load()
    dev = torch.device('cuda')
    torch.manual_seed(3)
    net = FlatMLP(17, 6, (256, 256), dev)
    for key, view in net.named_views():
        view.copy_(torch.randn(view.shape, device=dev) / (view.shape[-1] ** 0.5))
    M = 262144
    X = torch.zeros(M, 20, device=dev)
    X[:, :17] = torch.randn(M, 17, device=dev)
    net.forward(X, M, keep_acts=True)
    d = net.dout_view(M)
    d.copy_(torch.randn(d.shape, device=dev) * 0.01)

    dval = d.clone()

    def slabs():
        # (forward too: the first layer's streaming kernel keeps packed FMAs with the
        # swizzles that tested clean, and this holds them to it)
        out = net.forward(X, M, keep_acts=True).clone()
        acts = net._acts[:2 * M * 256].clone()
        d.copy_(dval)
        net.backward(X, M, d)
        return torch.cat([net._slabs.flatten(), out.flatten(), acts])

    ref = slabs()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    sink = torch.zeros(16, device=dev)
    burn = lib.ga_debug_mfma_burn
    burn.restype = C.c_int
    for mode in (2, 1):  # fp32 MFMAs, then bf16 MFMAs, on the other stream
        for _ in range(3):
            burn(C.c_int(mode), C.c_int(40000), C.c_int(512), C.c_void_p(sink.data_ptr()),
                 C.c_void_p(side.cuda_stream))
            got = slabs()
            torch.cuda.synchronize()
            assert torch.equal(got, ref), mode


def test_planes_written_by_the_optimizer_launch_are_the_plane_launchs(split_mode):
    """Inside an epoch call the reduction + Adam launch rewrites the bf16 planes of the
    weights it just updated, so that the next step's forward launch needs no plane
    launch (``ga_reduce_planes_hint``); with that off every forward launch computes
    them again.  Same planes, same bits."""
    import test_fused_train_gpu as T
    lib = split_mode
    spec, batch = T._problem('c3_shape')
    opt = (torch.optim.Adam, dict(lr=1e-3))
    res = []
    try:
        for on in (1, 0):
            lib.ga_set_split_adam_planes(on)
            algo, pol, vf = T._algo('c3_shape', spec, opt, epochs=2)
            np.random.seed(11)
            algo._train_once(0, batch)
            res.append((pol.net.params.clone(), vf.net.params.clone(),
                        pol.net.exp_avg_sq.clone(), dict(algo.last_tabular)))
    finally:
        lib.ga_set_split_adam_planes(1)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert torch.equal(res[0][2], res[1][2]) and res[0][3] == res[1][3]
