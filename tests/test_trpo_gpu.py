"""TRPO on the GPU: the constrained policy step against
goldens of the real ``garage.torch.algos.TRPO`` + ``ConjugateGradientOptimizer``
and against the oracle on a larger batch."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from test_ppo_gpu import LOG_KEYS, _host_batch, _sd, _spec

pytestmark = pytest.mark.gpu

TRPO_CASES = {
    'trpo': {},
    'trpo_tight': {},
    'trpo_reg': dict(entropy_method='regularized', policy_ent_coeff=0.02),
    'trpo_reject': {},
}


def _flat_no_pad(net, buf):
    """Flat vector in the reference's ``parameters()`` order (no padding)."""
    return np.concatenate([v.detach().cpu().numpy().reshape(-1)
                           for _, v in net.named_views(buf)])


def _make(g, tag, O, A, P, E, mb, **kw):
    from garage_amd.algos import TRPO
    from garage_amd.optimizers import (ConjugateGradientOptimizer,
                                       OptimizerWrapper)
    from garage_amd.policies import GaussianMLPPolicy, GaussianMLPValueFunction
    spec = _spec(O, A, P)
    pol = GaussianMLPPolicy(spec, hidden_sizes=(8, 8))
    vf = GaussianMLPValueFunction(spec, hidden_sizes=(8, 8))
    pol.load_state_dict(_sd(g, tag + '_pol0:'))
    vf.load_state_dict(_sd(g, tag + '_vf0:'))
    algo = TRPO(env_spec=spec, policy=pol, value_function=vf, sampler=None,
                policy_optimizer=OptimizerWrapper(
                    (ConjugateGradientOptimizer,
                     dict(max_constraint_value=float(g[tag + '_delta']),
                          max_backtracks=int(g[tag + '_max_backtracks']))),
                    pol),
                vf_optimizer=OptimizerWrapper(
                    (torch.optim.Adam, dict(lr=2.5e-4)), vf,
                    max_optimization_epochs=E, minibatch_size=mb),
                **kw)
    return spec, pol, vf, algo


@pytest.mark.parametrize('tag', sorted(TRPO_CASES))
def test_trpo_train_once_matches_real_reference(golden, tag):
    g = golden('trpo_train_once')
    O, A, P, E, mb = [int(v) for v in g[tag + '_cfg']]
    spec, pol, vf, algo = _make(g, tag, O, A, P, E, mb, **TRPO_CASES[tag])
    for it in range(2):
        pre = '%s_it%d_' % (tag, it)
        batch = _host_batch(spec, g, pre, O)
        np.random.seed(int(g[pre + 'np_seed']))
        algo._train_once(it, batch)
        cg = algo.last_cg
        # same tolerances as the oracle's own test against these goldens: ten
        # fp32 CG iterations amplify rounding to ~1e-4 of the direction, and the
        # second iteration starts from parameters that carry the first's
        grad = _flat_no_pad(pol.net, cg['grad'])
        assert np.allclose(grad, g[pre + 'cg:grad'],
                           atol=2e-6 if it == 0 else 5e-5)
        sd = _flat_no_pad(pol.net, cg['step_dir'])
        scale = np.abs(g[pre + 'cg:step_dir']).max()
        assert np.allclose(sd, g[pre + 'cg:step_dir'],
                           atol=(5e-4 if it == 0 else 4e-3) * scale)
        ds = _flat_no_pad(pol.net, cg['descent_step'])
        dscale = np.abs(g[pre + 'cg:descent_step']).max()
        assert np.allclose(ds, g[pre + 'cg:descent_step'],
                           atol=(1e-3 if it == 0 else 4e-3) * dscale)
        for mine, theirs in LOG_KEYS.items():
            want = float(g[pre + 'log:' + theirs])
            assert np.isclose(algo.last_tabular[mine], want,
                              atol=2e-5 if it == 0 else 2e-4,
                              rtol=1e-3), (mine, it, algo.last_tabular[mine],
                                           want)
        for k, v in pol.state_dict().items():
            assert np.allclose(v.numpy(), g[pre + 'pol:' + k],
                               atol=(1e-3 if it == 0 else 4e-3) * dscale), k
        for k, v in vf.state_dict().items():
            assert np.allclose(v.numpy(), g[pre + 'vf:' + k], atol=2e-6), k
    if tag == 'trpo_reject':
        pass  # first iteration rejected: parameters equal the golden's (above)


def _to_padded(net, flat):
    """The reference's flat ``parameters()``-order vector in the padded layout."""
    buf = torch.zeros(net.n_flat, dtype=torch.float32, device=net.device)
    off = 0
    for _, v in net.named_views(buf):
        n = v.numel()
        v.copy_(torch.from_numpy(
            np.ascontiguousarray(flat[off:off + n], dtype=np.float32)
        ).reshape(v.shape))
        off += n
    assert off == len(flat)
    return buf


@pytest.mark.parametrize('tag', sorted(TRPO_CASES))
def test_trpo_per_iterate_pins_against_real_reference(golden, tag):
    """Ten fp32 conjugate-gradient iterations amplify last-bit differences to
    ~1e-3 of the step (the end-to-end tolerance above; the reference run is itself
    one sample of that noise).  Here every operation of the constrained step is
    pinned on its own at the real reference's OWN iterates
    (``conjugate_gradient_optimizer.py:69-104,236-277``, recorded by
    ``make_golden.py``): ``A p_k`` for each of the ten directions the real
    ``_conjugate_gradient`` visited, and (loss, constraint) of every backtracking
    candidate of the real descent step -- each at one-operation fp32 accuracy."""
    g = golden('trpo_train_once')
    O, A, P, E, mb = [int(v) for v in g[tag + '_cfg']]
    spec, pol, vf, algo = _make(g, tag, O, A, P, E, mb, **TRPO_CASES[tag])
    pre = tag + '_it0_'
    batch = _host_batch(spec, g, pre, O)
    net = pol.net
    seen = {}
    real_train = algo._train

    def probing_train(dbatch, adv, returns, old_ll):
        M = dbatch.n_samples
        hyper = algo._policy_optimizer._hyper
        algo._trpo_share = 1.0
        # forward + loss at the step's starting point, as _train_policy does
        loss0, mean_old, dout = algo._policy_loss_pass(dbatch, adv, old_ll, M,
                                                       None, want_grad=True)
        mean_old = mean_old.clone()
        s_old = pol.clamped_log_std()
        z = torch.empty(net.n_flat, dtype=torch.float32, device=net.device)
        got = []
        for p_k in g[pre + 'cg:iter_p']:
            algo._fisher_vector_product(dbatch, M, _to_padded(net, p_k), z)
            got.append(_flat_no_pad(net, z))
        seen['Ap'] = np.stack(got)
        prev = net.params.clone()
        descent = _to_padded(net, g[pre + 'cg:descent_step'])
        ls = [float(loss0.item())]
        for k in range(len(g[pre + 'cg:ls_constraint'])):
            net.params.copy_(prev - float(hyper['backtrack_ratio'])**k * descent)
            l_new, mean_new, _ = algo._policy_loss_pass(dbatch, adv, old_ll, M,
                                                        None)
            kl = algo._kl_sum(mean_old, s_old, mean_new,
                              pol.clamped_log_std(), M)
            ls.append((float(l_new.item()), float(kl.item()) / M))
        net.params.copy_(prev)
        seen['ls'] = ls
        return real_train(dbatch, adv, returns, old_ll)

    algo._train = probing_train
    np.random.seed(int(g[pre + 'np_seed']))
    algo._train_once(0, batch)
    want = g[pre + 'cg:iter_Ap']
    assert seen['Ap'].shape == want.shape and len(want) == 10
    for k in range(len(want)):
        scale = np.abs(want[k]).max()
        assert np.allclose(seen['Ap'][k], want[k], atol=1e-5 * scale,
                           rtol=1e-5), (k, np.abs(seen['Ap'][k] - want[k]).max(),
                                        scale)
    ls = seen['ls']
    assert np.isclose(ls[0], g[pre + 'cg:ls_loss'][0], atol=2e-7)
    for k, (loss, kl) in enumerate(ls[1:]):
        assert np.isclose(loss, g[pre + 'cg:ls_loss'][k + 1], atol=1e-6,
                          rtol=1e-5), (k, loss)
        assert np.isclose(kl, g[pre + 'cg:ls_constraint'][k], atol=1e-7,
                          rtol=1e-4), (k, kl)


@pytest.mark.parametrize('options', ['default', 'layer_norm', 'relu_out_tanh'])
def test_fisher_vector_product_matches_double_backward(options):
    """``A v`` (tangent forward, Gaussian metric, backward) against the
    reference's Hessian-vector product by double backward (oracle) on the same
    parameters and observations -- also through a LayerNorm in front of every
    hidden layer (its tangent pass, ``ln_jvp_kernel``) and through relu hidden
    layers with a tanh on the mean."""
    from garage_amd._dtypes import Box, EnvSpec  # noqa: F401
    from garage_amd.algos import TRPO
    from garage_amd.policies import GaussianMLPPolicy, GaussianMLPValueFunction
    from oracle import networks as nets
    from oracle.trpo import build_hessian_vector_product
    O, A, P, M = 11, 3, 16, 700
    spec = _spec(O, A, P)
    torch.manual_seed(3)
    import contextlib
    pkw = {'default': {}, 'layer_norm': dict(layer_normalization=True),
           'relu_out_tanh': dict(hidden_nonlinearity=torch.relu,
                                 output_nonlinearity=torch.tanh)}[options]
    pol = GaussianMLPPolicy(spec, hidden_sizes=(64, 32), **pkw)
    if options == 'layer_norm':  # gamma / beta away from (1, 0)
        gen = torch.Generator(device='cpu').manual_seed(2)
        for name, view in pol.net.named_views():
            if 'layer_normalization' in name:
                view.add_(0.3 * torch.randn(view.shape, generator=gen).to(
                    view.device))
    vf = GaussianMLPValueFunction(spec, hidden_sizes=(8, ))
    algo = TRPO(env_spec=spec, policy=pol, value_function=vf, sampler=None)
    oracle_ctx = contextlib.ExitStack()
    if options == 'relu_out_tanh':
        oracle_ctx.enter_context(nets.hidden_nonlinearity(policy=torch.relu))
        oracle_ctx.enter_context(nets.output_nonlinearity(policy=torch.tanh))
    rng = np.random.RandomState(0)
    obs = rng.randn(M, O).astype(np.float32)
    from garage_amd.engine import pad_rows

    class B:
        obs_dev = pad_rows(obs)
        n_samples = M

    net = pol.net
    net.forward(B.obs_dev, M)  # activations at the current parameters
    # a tangent with zero padding, as every CG vector has
    vec = torch.zeros(net.n_flat, device=net.device)
    for _, v in net.named_views(vec):
        v.copy_(torch.from_numpy(rng.randn(*v.shape).astype(np.float32)))
    out = torch.empty_like(vec)
    algo._fisher_vector_product(B, M, vec, out)
    got = _flat_no_pad(net, out)
    # oracle: double backward through mean KL(old || new) at new == old
    params = OrderedDict((k, v.clone()) for k, v in pol.state_dict().items())
    old = OrderedDict((k, v.clone()) for k, v in params.items())
    keys = nets.trainable_keys(params)
    for k in keys:
        params[k].requires_grad_(True)
    x = torch.from_numpy(obs)

    def f_constraint():
        with torch.no_grad():
            d_old = nets.gaussian_dist(old, nets.POLICY_PREFIX, x)
        d_new = nets.gaussian_dist(params, nets.POLICY_PREFIX, x)
        return torch.distributions.kl.kl_divergence(d_old, d_new).mean()

    with oracle_ctx:
        f_Ax = build_hessian_vector_product(f_constraint,
                                            [params[k] for k in keys], 1e-5)
        v_ref = torch.from_numpy(_flat_no_pad(net, vec))
        want = f_Ax(v_ref).detach().numpy()
    assert np.allclose(got, want, atol=2e-5 * max(1.0, np.abs(want).max()),
                       rtol=1e-4)


# ---------------------------------------------------------------------------
# TRPO with the categorical head, against the real reference: TRPO +
# ConjugateGradientOptimizer on CategoricalCNNPolicy configured as an MLP
# (tests/_categorical_golden.py, tests/golden/trpo_categorical.npz)
TRPO_CATEGORICAL_CASES = {
    'trpo': {},
    'trpo3': {},
    'trpo_reg': dict(entropy_method='regularized', policy_ent_coeff=0.02),
    'trpo_c2': {},
}


def _views_no_std(net, buf):
    return [v for k, v in net.named_views(buf) if k != '_init_std']


def _flat_no_pad_cat(net, buf):
    return np.concatenate([v.detach().cpu().numpy().reshape(-1)
                           for v in _views_no_std(net, buf)])


def _to_padded_cat(net, flat):
    buf = torch.zeros(net.n_flat, dtype=torch.float32, device=net.device)
    off = 0
    for v in _views_no_std(net, buf):
        n = v.numel()
        v.copy_(torch.from_numpy(
            np.ascontiguousarray(flat[off:off + n], dtype=np.float32)
        ).reshape(v.shape))
        off += n
    assert off == len(flat)
    return buf


def _make_categorical(g, tag, **kw):
    import _categorical_golden as cg
    from garage_amd.algos import TRPO
    from garage_amd.optimizers import (ConjugateGradientOptimizer,
                                       OptimizerWrapper)
    from garage_amd.policies import (CategoricalMLPPolicy,
                                     GaussianMLPValueFunction)
    from test_ppo_gpu import _discrete_spec
    O, n_act, P, E, mb = [int(v) for v in g[tag + '_cfg']]
    hidden = tuple(int(h) for h in g[tag + '_hidden'])
    spec = _discrete_spec(O, n_act, P)
    pol = CategoricalMLPPolicy(spec, hidden_sizes=hidden)
    vf = GaussianMLPValueFunction(spec, hidden_sizes=hidden)
    pol.load_state_dict(cg.policy_params(g, tag + '_pol0:'))
    vf.load_state_dict(_sd(g, tag + '_vf0:'))
    algo = TRPO(env_spec=spec, policy=pol, value_function=vf, sampler=None,
                policy_optimizer=OptimizerWrapper(
                    (ConjugateGradientOptimizer,
                     dict(max_constraint_value=float(g[tag + '_delta']),
                          max_backtracks=int(g[tag + '_max_backtracks']))),
                    pol),
                vf_optimizer=OptimizerWrapper(
                    (torch.optim.Adam, dict(lr=2.5e-4)), vf,
                    max_optimization_epochs=E, minibatch_size=mb),
                **kw)
    return spec, pol, vf, algo, O


@pytest.mark.parametrize('tag', sorted(TRPO_CATEGORICAL_CASES))
def test_trpo_categorical_per_iterate_pins_against_real_reference(golden, tag):
    """The categorical metric (``ga_fisher_seed_categorical_f32``) inside
    ``A p_k = J^T M J p_k + reg p_k`` for each of the ten directions the real
    ``_conjugate_gradient`` visited on the reference's categorical policy, and
    (loss, constraint) of every backtracking candidate of its real descent step
    -- each at one-operation fp32 accuracy (the end-to-end comparison of a
    softmax head is loose by nature, see the test below)."""
    g = golden('trpo_categorical')
    spec, pol, vf, algo, O = _make_categorical(
        g, tag, **TRPO_CATEGORICAL_CASES[tag])
    pre = tag + '_it0_'
    batch = _host_batch(spec, g, pre, O)
    net = pol.net
    seen = {}
    real_train = algo._train

    def probing_train(dbatch, adv, returns, old_ll):
        M = dbatch.n_samples
        hyper = algo._policy_optimizer._hyper
        algo._trpo_share = 1.0
        loss0, head_old, dout = algo._policy_loss_pass(dbatch, adv, old_ll, M,
                                                       None, want_grad=True)
        head_old = head_old.clone()
        z = torch.empty(net.n_flat, dtype=torch.float32, device=net.device)
        got = []
        for p_k in g[pre + 'cg:iter_p']:
            algo._fisher_vector_product(dbatch, M, _to_padded_cat(net, p_k), z)
            got.append(_flat_no_pad_cat(net, z))
        seen['Ap'] = np.stack(got)
        prev = net.params.clone()
        descent = _to_padded_cat(net, g[pre + 'cg:descent_step'])
        ls = [float(loss0.item())]
        for k in range(len(g[pre + 'cg:ls_constraint'])):
            net.params.copy_(prev - float(hyper['backtrack_ratio'])**k * descent)
            l_new, head_new, _ = algo._policy_loss_pass(dbatch, adv, old_ll, M,
                                                        None)
            kl = algo._kl_sum(head_old, 0.0, head_new, 0.0, M)
            ls.append((float(l_new.item()), float(kl.item()) / M))
        net.params.copy_(prev)
        seen['ls'] = ls
        return real_train(dbatch, adv, returns, old_ll)

    algo._train = probing_train
    np.random.seed(int(g[pre + 'np_seed']))
    algo._train_once(0, batch)
    want = g[pre + 'cg:iter_Ap']
    assert seen['Ap'].shape == want.shape and len(want) == 10
    for k in range(len(want)):
        scale = np.abs(want[k]).max()
        assert np.allclose(seen['Ap'][k], want[k], atol=1e-5 * scale,
                           rtol=1e-5), (k, np.abs(seen['Ap'][k] - want[k]).max(),
                                        scale)
    ls = seen['ls']
    assert np.isclose(ls[0], g[pre + 'cg:ls_loss'][0], atol=2e-7)
    for k, (loss, kl) in enumerate(ls[1:]):
        assert np.isclose(loss, g[pre + 'cg:ls_loss'][k + 1], atol=1e-6,
                          rtol=1e-5), (k, loss)
        assert np.isclose(kl, g[pre + 'cg:ls_constraint'][k], atol=1e-7,
                          rtol=1e-4), (k, kl)


@pytest.mark.parametrize('tag', sorted(TRPO_CATEGORICAL_CASES))
def test_trpo_categorical_train_once_matches_real_reference(golden, tag):
    """Two whole iterations.  Loose by nature (the tolerances of the oracle's own
    test against this fixture, ``test_golden_trpo_categorical_train_once``): the
    KL Hessian of a softmax head has the null direction "every score + c", kept
    finite only by ``hvp_reg_coeff``, and ten fp32 CG iterations amplify last-bit
    differences to 0.05-2 % of the direction; the starting gradient and the
    number of backtracking candidates are tight."""
    import _categorical_golden as cg
    g = golden('trpo_categorical')
    spec, pol, vf, algo, O = _make_categorical(
        g, tag, **TRPO_CATEGORICAL_CASES[tag])
    for it in range(2):
        pre = '%s_it%d_' % (tag, it)
        batch = _host_batch(spec, g, pre, O)
        np.random.seed(int(g[pre + 'np_seed']))
        algo._train_once(it, batch)
        last = algo.last_cg
        if it == 0:
            grad = _flat_no_pad_cat(pol.net, last['grad'])
            assert np.allclose(grad, g[pre + 'cg:grad'], atol=2e-6)
            assert last['accepted'] + 1 == len(g[pre + 'cg:ls_constraint'])
        loose = 3e-2 if it == 0 else 6e-2
        sd = _flat_no_pad_cat(pol.net, last['step_dir'])
        scale = np.abs(g[pre + 'cg:step_dir']).max()
        assert np.allclose(sd, g[pre + 'cg:step_dir'], atol=loose * scale)
        ds = _flat_no_pad_cat(pol.net, last['descent_step'])
        dscale = np.abs(g[pre + 'cg:descent_step']).max()
        assert np.allclose(ds, g[pre + 'cg:descent_step'], atol=loose * dscale)
        for mine, theirs in cg.LOG_KEYS.items():
            want = float(g[pre + 'log:' + theirs])
            assert np.isclose(algo.last_tabular[mine], want,
                              atol=2e-4 if it == 0 else 1e-3,
                              rtol=5e-2), (mine, it, algo.last_tabular[mine],
                                           want)
        want_pol = cg.policy_params(g, pre + 'pol:')
        for k, v in pol.state_dict().items():
            assert np.allclose(v.numpy(), want_pol[k].numpy(),
                               atol=loose * dscale), k
        for k, v in vf.state_dict().items():
            assert np.allclose(v.numpy(), g[pre + 'vf:' + k], atol=2e-6), k


@pytest.mark.parametrize('double_softmax', [True, False])
def test_categorical_fisher_vector_product_matches_double_backward(
        double_softmax):
    """``A v`` through the categorical metric against the reference's
    Hessian-vector product by double backward (oracle) at 5 classes, with and
    without the head's inner softmax."""
    from garage_amd.algos import TRPO
    from garage_amd.engine import pad_rows
    from garage_amd.policies import (CategoricalMLPPolicy,
                                     GaussianMLPValueFunction)
    from oracle import networks as nets
    from oracle.trpo import build_hessian_vector_product
    from test_ppo_gpu import _discrete_spec
    O, n_act, P, M = 11, 5, 16, 700
    spec = _discrete_spec(O, n_act, P)
    torch.manual_seed(4)
    pol = CategoricalMLPPolicy(spec, hidden_sizes=(64, 32),
                               double_softmax=double_softmax)
    gen = torch.Generator(device='cpu').manual_seed(6)
    for name, view in pol.net.named_views():
        if name != '_init_std':  # away from the zero-bias init
            view.add_(0.3 * torch.randn(view.shape, generator=gen).to(
                view.device))
    vf = GaussianMLPValueFunction(spec, hidden_sizes=(8, ))
    algo = TRPO(env_spec=spec, policy=pol, value_function=vf, sampler=None)
    rng = np.random.RandomState(0)
    obs = rng.randn(M, O).astype(np.float32)

    class B:
        obs_dev = pad_rows(obs)
        n_samples = M

    net = pol.net
    net.forward(B.obs_dev, M)  # activations + scores at the current parameters
    vec = torch.zeros(net.n_flat, device=net.device)
    for v in _views_no_std(net, vec):
        v.copy_(torch.from_numpy(rng.randn(*v.shape).astype(np.float32)))
    out = torch.empty_like(vec)
    algo._fisher_vector_product(B, M, vec, out)
    got = _flat_no_pad_cat(net, out)
    assert float(out[0]) == 0.0  # the layout's std slot is not a parameter
    params = OrderedDict((k, v.clone()) for k, v in pol.state_dict().items())
    old = OrderedDict((k, v.clone()) for k, v in params.items())
    keys = nets.trainable_keys(params)
    for k in keys:
        params[k].requires_grad_(True)
    x = torch.from_numpy(obs)

    def f_constraint():
        with torch.no_grad():
            d_old = nets.categorical_dist(old, nets.POLICY_PREFIX, x,
                                          double_softmax)
        d_new = nets.categorical_dist(params, nets.POLICY_PREFIX, x,
                                      double_softmax)
        return torch.distributions.kl.kl_divergence(d_old, d_new).mean()

    f_Ax = build_hessian_vector_product(f_constraint,
                                        [params[k] for k in keys], 1e-5)
    want = f_Ax(torch.from_numpy(_flat_no_pad_cat(net, vec))).detach().numpy()
    assert np.allclose(got, want, atol=2e-5 * max(1.0, np.abs(want).max()),
                       rtol=1e-4)


def test_trpo_iteration_matches_oracle_larger_batch():
    """One TRPO iteration on ~3000 samples, 2 hidden layers of 64: post-step
    parameters, logged scalars and the accepted backtracking index against the
    oracle (torch CPU double backward)."""
    from garage_amd._dtypes import EpisodeBatch, StepType
    from garage_amd.algos import TRPO
    from garage_amd.optimizers import (ConjugateGradientOptimizer,
                                       OptimizerWrapper)
    from garage_amd.policies import GaussianMLPPolicy, GaussianMLPValueFunction
    from oracle import batch as ob
    from oracle.trpo import OracleTRPO
    O, A, P = 9, 4, 32
    spec = _spec(O, A, P)
    torch.manual_seed(5)
    rng = np.random.RandomState(5)
    pol = GaussianMLPPolicy(spec, hidden_sizes=(64, 64))
    vf = GaussianMLPValueFunction(spec, hidden_sizes=(64, 64))
    lens = rng.randint(5, P + 1, size=150)
    lens[::7] = P
    S = int(lens.sum())
    st = []
    for L in lens:
        t = [1] * L
        t[0] = 0
        t[-1] = 3 if L == P else 2
        st += t
    obs = rng.randn(S, O).astype(np.float32)
    acts = rng.randn(S, A).astype(np.float32)
    rew = rng.randn(S)
    oracle = OracleTRPO(OrderedDict(pol.state_dict()),
                        OrderedDict(vf.state_dict()), max_episode_length=P,
                        max_optimization_epochs=2, minibatch_size=512)
    algo = TRPO(env_spec=spec, policy=pol, value_function=vf, sampler=None,
                policy_optimizer=OptimizerWrapper(
                    (ConjugateGradientOptimizer,
                     dict(max_constraint_value=0.01)), pol),
                vf_optimizer=OptimizerWrapper(
                    (torch.optim.Adam, dict(lr=2.5e-4)), vf,
                    max_optimization_epochs=2, minibatch_size=512))
    b = ob.OracleEpisodeBatch(observations=obs,
                              last_observations=np.zeros((len(lens), O),
                                                         np.float32),
                              actions=acts, rewards=rew,
                              step_types=np.asarray(st), lengths=lens,
                              max_episode_length=P)
    np.random.seed(9)
    want = oracle.train_once(b)
    batch = EpisodeBatch(env_spec=spec, episode_infos={}, observations=obs,
                         last_observations=np.zeros((len(lens), O), np.float32),
                         actions=acts, rewards=rew, env_infos={},
                         agent_infos={},
                         step_types=np.asarray([StepType(s) for s in st],
                                               dtype=object),
                         lengths=lens.astype('l'))
    np.random.seed(9)
    algo._train_once(0, batch)
    assert algo.last_cg['accepted'] == oracle.cg.trace['accepted']
    dscale = np.abs(oracle.cg.trace['descent_step']).max()
    ds = _flat_no_pad(pol.net, algo.last_cg['descent_step'])
    assert np.allclose(ds, oracle.cg.trace['descent_step'], atol=2e-3 * dscale)
    for k in ('policy/LossBefore', 'policy/LossAfter', 'policy/KL',
              'policy/Entropy', 'vf/LossBefore', 'vf/LossAfter'):
        assert np.isclose(algo.last_tabular[k], want[k], atol=2e-5,
                          rtol=1e-3), (k, algo.last_tabular[k], want[k])
    wpol, wvf = oracle.state()
    for k, v in pol.state_dict().items():
        assert np.allclose(v.numpy(), wpol[k], atol=2e-3 * dscale), k
    for k, v in vf.state_dict().items():
        assert np.allclose(v.numpy(), wvf[k], atol=5e-6), k


def test_trpo_overlapped_value_passes_equal_serial_bitwise():
    """TRPO on one GPU runs the value function's epochs on a second stream under
    the policy step (``garage_amd.algos.TRPO._train``); the reference finishes the
    policy first (``vpg.py:244-248``).  The two schedules must hold the same bits:
    at C3's size (1 M samples, E = 10 x 32 minibatches, device permutations) the
    side stream lags the host by many epochs, so a permutation that is drawn
    lazily, or whose memory is recycled while the side stream still gathers
    through it, shows up here as a difference (round-2 advisor finding)."""
    import bench
    cfg = bench.CONFIGS['c3']
    algo, sampler, pol, S = bench.build_engine(cfg, None, seed=2,
                                               algo_name='trpo')
    vf = algo._value_function
    eps = sampler.obtain_samples(0, S, None)

    def snap():
        return (pol.net.params.clone(), vf.net.params.clone(),
                vf.net.exp_avg.clone(), vf.net.exp_avg_sq.clone(),
                vf.net.adam_steps, algo._vf_optimizer._draws,
                algo._old_policy.params.clone())

    def restore(s):
        pol.net.params.copy_(s[0])
        vf.net.params.copy_(s[1])
        vf.net.exp_avg.copy_(s[2])
        vf.net.exp_avg_sq.copy_(s[3])
        vf.net.adam_steps = s[4]
        algo._vf_optimizer._draws = s[5]
        algo._old_policy.params.copy_(s[6])

    s0 = snap()
    results = []
    for overlap in (True, False, True):
        restore(s0)
        algo.overlap_updates = overlap
        algo._train_once(0, eps)
        torch.cuda.synchronize()
        results.append((pol.net.params.clone(), vf.net.params.clone(),
                        vf.net.exp_avg_sq.clone(), dict(algo.last_tabular)))
    for got in results[1:]:
        assert torch.equal(got[0], results[0][0])
        assert torch.equal(got[1], results[0][1])
        assert torch.equal(got[2], results[0][2])
        assert got[3] == results[0][3]
    assert not torch.equal(results[0][1], s0[1])
