"""The fused optimizer-step kernels (fused_train.hip) against the per-layer path.

``ga_update_epoch*`` takes the fused kernels by default when the network's last
hidden layer is 64 / 128 / 256 units wide: last hidden layer + head + loss +
gradient seed + head weight gradient in one launch, data gradient into the first
hidden layer + first-layer weight gradient in one launch, one reduction + Adam
launch (``VPG._train_policy`` / ``_train_value_function``, torch/algos/vpg.py
:250-293).  Oracle / golden parity of that default path is what every iteration
test of this suite checks (test_ppo_gpu, test_configs_gpu, test_kernels_gpu ...);
here the two paths of the library are held against each other gradient by
gradient, on shapes that exercise every branch: each tile width, one to three
hidden layers, Gaussian / categorical / value heads, VPG and PPO objectives,
entropy terms, ragged last tiles, gathered rows.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = {
    # O, A, hidden, minibatch, discrete, algo keywords
    'c3_shape': (17, 6, (256, 256), 4096, False, {}),
    'c2_shape_categorical': (4, 2, (64, 64), 1024, True, {}),
    'ragged_tiles_128': (9, 3, (128, 128), 1000, False, {}),
    'three_hidden': (11, 5, (128, 64, 128), 777, False, {}),
    'one_hidden': (12, 4, (128, ), 640, False, {}),
    'wide_first_narrow_last': (7, 8, (96, 64), 513, False, {}),
    'wide_input': (40, 2, (64, 256), 900, False, {}),
    'vpg': (6, 3, (64, 64), 333, False, {'vpg': True}),
    'entropy_regularized': (6, 3, (128, 256), 500, False,
                            {'entropy_method': 'regularized',
                             'policy_ent_coeff': 0.02,
                             'use_softplus_entropy': True}),
    'categorical_max_entropy': (5, 4, (64, 128), 450, True,
                                {'entropy_method': 'max',
                                 'policy_ent_coeff': 0.01, 'center_adv': False,
                                 'stop_entropy_gradient': True}),
    'full_batch': (8, 2, (64, 64), None, False, {}),
    # 17 .. 20 inputs at 256 units: the software-pipelined k-loop instantiation
    # (c3_shape is one; these add the other input widths, ragged last tiles --
    # the predicated-store variant -- and a one-output head)
    'pipelined_obs18_ragged': (18, 6, (256, 256), 1000, False, {}),
    'pipelined_obs19_vpg': (19, 3, (256, 256), 777, False, {'vpg': True}),
    'pipelined_obs20_categorical': (20, 5, (256, 256), 2048, True, {}),
    # 2 x 32 / 2 x 64 networks: the whole step in one launch (narrow_step.hip)
    'narrow_32': (4, 2, (32, 32), 200, True, {}),
    'narrow_32_gaussian_wide_input': (32, 8, (32, 32), 333, False, {}),
    'narrow_64_ragged': (13, 5, (64, 64), 1000, False, {}),
    'narrow_64_entropy': (6, 3, (64, 64), 450, False,
                          {'entropy_method': 'regularized',
                           'policy_ent_coeff': 0.03}),
}


def _problem(case):
    from garage_amd._dtypes import Box, Discrete, EnvSpec, EpisodeBatch, StepType
    O, A, hidden, mb, discrete, kw = CASES[case]
    P = 40
    act_space = Discrete(A) if discrete else Box(-np.inf, np.inf, (A, ))
    spec = EnvSpec(Box(-np.inf, np.inf, (O, )), act_space, max_episode_length=P)
    rng = np.random.RandomState(len(case))
    lens = rng.randint(5, P + 1, size=230)
    lens[0] = P
    S = int(lens.sum())
    st = []
    for n in lens:
        t = [1] * n
        t[0] = 0
        t[-1] = 3 if n == P else 2
        st += t
    acts = (rng.randint(0, A, size=S).astype(np.int64) if discrete else
            rng.randn(S, A).astype(np.float32))
    batch = EpisodeBatch(env_spec=spec, episode_infos={},
                         observations=rng.randn(S, O).astype(np.float32),
                         last_observations=np.zeros((len(lens), O), np.float32),
                         actions=acts, rewards=rng.randn(S), env_infos={},
                         agent_infos={},
                         step_types=np.asarray([StepType(s) for s in st],
                                               dtype=object),
                         lengths=lens.astype('l'))
    return spec, batch


def _algo(case, spec, opt, epochs=1):
    from garage_amd.algos import PPO, VPG
    from garage_amd.optimizers import OptimizerWrapper
    from garage_amd.policies import (CategoricalMLPPolicy, GaussianMLPPolicy,
                                     GaussianMLPValueFunction)
    O, A, hidden, mb, discrete, kw = CASES[case]
    kw = dict(kw)
    cls = VPG if kw.pop('vpg', False) else PPO
    torch.manual_seed(3)
    pol_cls = CategoricalMLPPolicy if discrete else GaussianMLPPolicy
    pol = pol_cls(spec, hidden_sizes=hidden)
    vf = GaussianMLPValueFunction(spec, hidden_sizes=hidden)
    # biases and the log-std away from their zero initialisation
    for net in (pol.net, vf.net):
        g = torch.Generator(device='cpu').manual_seed(5)
        for l in range(len(net.dims) - 1):
            net.bias(l).copy_(0.1 * torch.randn(net.dims[l + 1], generator=g))
        if net is vf.net or not discrete:
            net.params[0] = -0.3
    algo = cls(env_spec=spec, policy=pol, value_function=vf, sampler=None,
               policy_optimizer=OptimizerWrapper(opt, pol, epochs, mb),
               vf_optimizer=OptimizerWrapper(opt, vf, epochs, mb), **kw)
    return algo, pol, vf


@pytest.mark.parametrize('case', sorted(CASES))
def test_fused_step_gradients_match_per_layer_path(case):
    """Adam with beta1 = beta2 = 0 and eps = 1 moves a parameter by
    -lr g / (|g| + 1): (new - old) / lr exposes the gradients of every step of a
    pass over the batch (several minibatches, the last one ragged)."""
    from garage_amd import _lib
    lib = _lib.load()
    spec, batch = _problem(case)
    lr = 1e-3
    out = {}
    try:
        for on in (1, 0):
            lib.ga_set_fused_train(on)
            lib.ga_set_small_step(on)  # reference path: per-layer kernels only
            algo, pol, vf = _algo(
                case, spec, (torch.optim.Adam, dict(lr=lr, betas=(0.0, 0.0),
                                                    eps=1.0)))
            p0, v0 = pol.net.params.clone(), vf.net.params.clone()
            np.random.seed(11)
            algo._train_once(0, batch)
            out[on] = ((pol.net.params - p0) / lr, (vf.net.params - v0) / lr,
                       dict(algo.last_tabular))
    finally:
        lib.ga_set_fused_train(1)
        lib.ga_set_small_step(1)
    for i in (0, 1):
        scale = float(out[0][i].abs().max())
        assert scale > 1e-3  # the pass did move the parameters
        d = float((out[1][i] - out[0][i]).abs().max())
        assert d < 1e-4 * scale + 4e-5, (i, d, scale)
    for k, v in out[0][2].items():
        assert np.isclose(out[1][2][k], v, rtol=2e-5, atol=2e-6), (k, v,
                                                                  out[1][2][k])


@pytest.mark.parametrize('case', ['c3_shape', 'c2_shape_categorical',
                                  'three_hidden', 'narrow_32'])
def test_fused_step_python_loop_and_native_loop_are_the_same_bits(case):
    """A subclass that hooks ``_train_policy`` forces the Python minibatch loop,
    which drives the same entry point one minibatch at a time: same bits as the
    C++ epoch loop, on one stream and on two."""
    from garage_amd.algos import PPO
    spec, batch = _problem(case)
    opt = (torch.optim.Adam, dict(lr=1e-3))
    res = []
    for mode in ('pair', 'serial', 'python'):
        algo, pol, vf = _algo(case, spec, opt, epochs=2)
        if mode == 'python':
            def train_policy(*args, _orig=algo._train_policy):
                return _orig(*args)
            # instance attribute is not enough for _native_update_ok: subclass
            algo.__class__ = type('Hooked', (algo.__class__, ), {
                '_train_policy': lambda self, *a: PPO._train_policy(self, *a)})
            assert not algo._native_update_ok()
        algo.overlap_updates = mode == 'pair'
        np.random.seed(11)
        algo._train_once(0, batch)
        res.append((pol.net.params.clone(), vf.net.params.clone(),
                    dict(algo.last_tabular)))
    for other in res[1:]:
        assert torch.equal(res[0][0], other[0])
        assert torch.equal(res[0][1], other[1])
        assert res[0][2] == other[2]


def test_fused_step_is_reproducible_and_actually_taken():
    """Two runs from the same state give the same bits, and switching the fused
    kernels off changes the bits (i.e. the default path really is the fused one)."""
    from garage_amd import _lib
    lib = _lib.load()
    spec, batch = _problem('c3_shape')
    opt = (torch.optim.Adam, dict(lr=1e-3))
    res = []
    try:
        for on in (1, 1, 0):
            lib.ga_set_fused_train(on)
            algo, pol, vf = _algo('c3_shape', spec, opt, epochs=2)
            np.random.seed(11)
            algo._train_once(0, batch)
            res.append((pol.net.params.clone(), vf.net.params.clone()))
    finally:
        lib.ga_set_fused_train(1)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert not torch.equal(res[0][0], res[2][0])
    assert float((res[0][0] - res[2][0]).abs().max()) < 5e-4


def test_narrow_step_is_taken_and_matches_the_gemm_epilogue_kernels():
    """2 x 64 networks are eligible for both fused paths: the one-launch narrow
    step is the default; switching it off falls back to the GEMM-epilogue kernels;
    all three (with the per-layer kernels) agree to rounding."""
    from garage_amd import _lib
    lib = _lib.load()
    spec, batch = _problem('c2_shape_categorical')
    opt = (torch.optim.Adam, dict(lr=1e-3, betas=(0.0, 0.0), eps=1.0))
    res = []
    try:
        for narrow, fused in ((1, 1), (0, 1), (0, 0)):
            lib.ga_set_narrow_step(narrow)
            lib.ga_set_fused_train(fused)
            algo, pol, vf = _algo('c2_shape_categorical', spec, opt)
            p0, v0 = pol.net.params.clone(), vf.net.params.clone()
            np.random.seed(11)
            algo._train_once(0, batch)
            res.append(((pol.net.params - p0) / 1e-3, (vf.net.params - v0) / 1e-3))
    finally:
        lib.ga_set_narrow_step(1)
        lib.ga_set_fused_train(1)
    assert not torch.equal(res[0][0], res[1][0])  # different kernels ran
    for other in res[1:]:
        for i in (0, 1):
            scale = float(res[2][i].abs().max())
            d = float((res[0][i] - other[i]).abs().max())
            assert d < 1e-4 * scale + 4e-5, (i, d, scale)


@pytest.mark.parametrize('case', ['c3_shape', 'ragged_tiles_128',
                                  'wide_first_narrow_last',
                                  'categorical_max_entropy'])
def test_first_layer_inside_the_fused_kernel_is_taken_and_matches(case):
    """Two hidden layers and <= 32 inputs: the last-hidden-layer kernel produces the
    first layer's outputs itself (v_mfma_f32_16x16x4_f32 sub-tiles per k-chunk);
    ``ga_set_fused_first_layer(0)`` restores the separate streaming launch.  Other
    summation order in the first layer, same gradients to rounding."""
    from garage_amd import _lib
    lib = _lib.load()
    spec, batch = _problem(case)
    lr = 1e-3
    opt = (torch.optim.Adam, dict(lr=lr, betas=(0.0, 0.0), eps=1.0))
    res = []
    try:
        for on in (1, 0):
            lib.ga_set_fused_first_layer(on)
            algo, pol, vf = _algo(case, spec, opt)
            p0, v0 = pol.net.params.clone(), vf.net.params.clone()
            np.random.seed(11)
            algo._train_once(0, batch)
            res.append(((pol.net.params - p0) / lr, (vf.net.params - v0) / lr,
                        dict(algo.last_tabular)))
    finally:
        lib.ga_set_fused_first_layer(1)
    assert not torch.equal(res[0][0], res[1][0])  # different kernels ran
    for i in (0, 1):
        scale = float(res[1][i].abs().max())
        d = float((res[0][i] - res[1][i]).abs().max())
        assert d < 1e-4 * scale + 4e-5, (i, d, scale)
    for k, v in res[1][2].items():
        assert np.isclose(res[0][2][k], v, rtol=2e-5, atol=2e-6), (k, v,
                                                                  res[0][2][k])


@pytest.mark.parametrize('shape', [
    # in, hidden, out, rows
    (17, (256, 256), 6, 70000),
    (17, (256, 256), 1, 4097),
    (9, (128, 128), 3, 8191),
    (32, (128, 256), 8, 4096),
    (20, (256, 128), 2, 6000),
])
def test_outputs_only_forward_matches_the_per_layer_forward(shape):
    """``forward(keep_acts=False)`` -- the full-batch evaluation passes of
    ``_train_once`` (vpg.py:147-184: baselines, old log-likelihoods, LossBefore /
    LossAfter / KL) -- computes the whole network in one launch
    (``mlp_eval_forward_kernel``) and writes only the outputs: same values as the
    per-layer kernels to rounding, gathered rows and ragged last tiles included,
    and the activation workspace of an earlier pass is left alone."""
    from garage_amd import _lib
    from garage_amd.engine import FlatMLP
    in_dim, hidden, out_dim, M = shape
    lib = _lib.load()
    dev = torch.device('cuda:0')
    torch.manual_seed(3)
    net = FlatMLP(in_dim, out_dim, hidden, dev)
    net.params.copy_(torch.randn_like(net.params) * 0.3)
    import ctypes as C
    assert lib.ga_mlp_forward_eval_supported(C.byref(net._desc)) == 1
    rows = M + 1000
    X = torch.randn(rows, (in_dim + 3) // 4 * 4, device=dev)
    idx = torch.randperm(rows, device=dev)[:M].to(torch.int32)
    for row_idx in (None, idx):
        ref = net.forward(X, M, row_idx=row_idx).clone()
        acts = net._acts.clone()
        net._acts.fill_(7.0)
        got = net.forward(X, M, row_idx=row_idx, keep_acts=False).clone()
        assert torch.equal(net._acts, torch.full_like(net._acts, 7.0))
        net._acts.copy_(acts)
        assert torch.isfinite(got[:, :out_dim]).all()
        assert torch.allclose(got[:, :out_dim], ref[:, :out_dim], atol=2e-5,
                              rtol=1e-5)
    # networks outside the kernel's shapes keep the per-layer path
    other = FlatMLP(40, 2, (128, 128), dev)
    assert lib.ga_mlp_forward_eval_supported(C.byref(other._desc)) == 0
    narrow = FlatMLP(4, 2, (64, 64), dev)  # the 64 x 64 GEMM tiles are faster there
    assert lib.ga_mlp_forward_eval_supported(C.byref(narrow._desc)) == 0
    relu = FlatMLP(in_dim, out_dim, hidden, dev, hidden_act='relu')
    assert lib.ga_mlp_forward_eval_supported(C.byref(relu._desc)) == 0
    y = relu.forward(X, M, keep_acts=False)
    assert torch.isfinite(y[:, :out_dim]).all()


@pytest.mark.parametrize('case', ['c3_shape', 'pipelined_obs18_ragged',
                                  'pipelined_obs20_categorical'])
def test_pipelined_kloop_is_bit_identical_to_the_plain_loop(case):
    """``fwd_head_loss_kernel<256,1,8,true,5>`` / ``mlp_eval_forward_kernel<256,1,8,5>``
    issue the first-layer producer, the H1 spill and the weight prefetch between the
    MFMAs of a k-step (round 3); every output element keeps its k order, so a whole
    iteration -- several minibatches, ragged last tiles, the full-batch evaluation
    passes -- must give the same BITS as the plain loop (``ga_set_pipelined_kloop(0)``)."""
    from garage_amd import _lib
    lib = _lib.load()
    spec, batch = _problem(case)
    opt = (torch.optim.Adam, dict(lr=1e-3))
    res = []
    try:
        for on in (1, 0):
            lib.ga_set_pipelined_kloop(on)
            algo, pol, vf = _algo(case, spec, opt, epochs=2)
            np.random.seed(11)
            algo._train_once(0, batch)
            res.append((pol.net.params.clone(), vf.net.params.clone(),
                        pol.net.exp_avg_sq.clone(), dict(algo.last_tabular)))
    finally:
        lib.ga_set_pipelined_kloop(1)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert torch.equal(res[0][2], res[1][2]) and res[0][3] == res[1][3]
