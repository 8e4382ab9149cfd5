"""Gaussian MLP policy and value function on flat HBM parameter buffers.

API-compatible with ``garage.torch.policies.GaussianMLPPolicy``
(``torch/policies/gaussian_mlp_policy.py:9-102``, ``stochastic_policy.py:11-104``,
``torch/policies/policy.py:9-79``) and
``garage.torch.value_functions.GaussianMLPValueFunction``
(``torch/value_functions/gaussian_mlp_value_function.py:9-112``): same
constructor keywords, ``state_dict`` key names and shapes, ``get_actions`` /
``get_param_values`` / ``set_param_values`` / ``forward`` / ``compute_loss``.
The arithmetic runs in the HIP kernels of ``garage_amd/csrc``.
"""
import ctypes as C
import math
from collections import OrderedDict

import numpy as np
import torch
from torch import nn
from torch.distributions import Independent, Normal

from garage_amd import _lib
from garage_amd._dtypes import is_discrete
from garage_amd._lib import call, dptr, stream_ptr
from garage_amd.engine import (FlatMLP, pad_rows, reduction_workspace,
                               require_gpu, round4)


def _hidden_act(hidden_nonlinearity):
    """``hidden_nonlinearity`` of the reference's MLP modules
    (``torch/modules/mlp_module.py:43-44``, wrapped by ``NonLinearity``,
    ``multi_headed_mlp_module.py:154-197``: a callable or an ``nn.Module``
    class / instance, ``None`` = linear) as the kernels' activation name."""
    import torch.nn.functional as F
    h = hidden_nonlinearity
    if h is None:
        return 'none'
    if h in (torch.tanh, 'tanh', nn.Tanh, F.tanh) or isinstance(h, nn.Tanh):
        return 'tanh'
    if h in (torch.relu, 'relu', nn.ReLU, F.relu) or isinstance(h, nn.ReLU):
        return 'relu'
    if h in (torch.sigmoid, 'sigmoid', nn.Sigmoid, F.sigmoid) or \
            isinstance(h, nn.Sigmoid):
        return 'sigmoid'
    # parameterised activations: the kernels implement torch's DEFAULT settings
    # (elu alpha = 1, leaky_relu negative_slope = 0.01, softplus beta = 1 /
    # threshold = 20); the functional forms and the module CLASSES mean those
    if h in (F.elu, 'elu', nn.ELU) or (isinstance(h, nn.ELU)
                                      and h.alpha == 1.0):
        return 'elu'
    if h in (F.leaky_relu, 'leaky_relu', nn.LeakyReLU) or (
            isinstance(h, nn.LeakyReLU) and h.negative_slope == 0.01):
        return 'leaky_relu'
    if h in (F.softplus, 'softplus', nn.Softplus) or (
            isinstance(h, nn.Softplus) and h.beta in (1, 1.0)
            and h.threshold in (20, 20.0)):
        return 'softplus'
    raise NotImplementedError(
        'garage_amd kernels implement tanh (the GaussianMLP* default), relu, '
        'sigmoid, elu, leaky_relu, softplus (torch defaults) and linear layers '
        '-- activations whose slope is a function of their output; got '
        '{!r}'.format(hidden_nonlinearity))


def _check_supported(hidden_nonlinearity, output_nonlinearity,
                     std_parameterization, layer_normalization):
    """-> (hidden activation, output activation, layer norm) for FlatMLP."""
    act = (_hidden_act(hidden_nonlinearity), _hidden_act(output_nonlinearity),
           bool(layer_normalization))
    if std_parameterization not in ('exp', 'softplus'):
        raise NotImplementedError  # gaussian_mlp_module.py:120-121
    return act


def _reference_init(mlp, hidden_w_init, hidden_b_init, output_w_init,
                    output_b_init):
    """Initialise exactly as the reference does, consuming the same torch RNG.

    ``MultiHeadedMLPModule.__init__`` (multi_headed_mlp_module.py:70-101)
    builds an ``nn.Linear`` (which draws its own default init) and then applies
    the weight / bias initialisers; doing the same on the CPU and copying the
    result means ``set_seed(s)`` + construction yields the reference's
    parameters bit for bit.
    """
    nl = len(mlp.dims) - 1
    for l in range(nl):
        lin = nn.Linear(mlp.dims[l], mlp.dims[l + 1])
        if l < nl - 1:
            hidden_w_init(lin.weight)
            hidden_b_init(lin.bias)
        else:
            output_w_init(lin.weight)
            output_b_init(lin.bias)
        mlp.weight(l).copy_(lin.weight.detach())
        mlp.bias(l).copy_(lin.bias.detach())


class _GaussianMLP:
    """Shared machinery of the policy and the value function."""

    _prefix = '_module.'

    def _build(self, in_dim, out_dim, hidden_sizes, hidden_w_init,
               hidden_b_init, output_w_init, output_b_init, learn_std,
               init_std, min_std, max_std, device, hidden_act='tanh'):
        self.device = device or require_gpu()
        output_act, layer_norm = 'none', False
        if isinstance(hidden_act, tuple):
            hidden_act, output_act, layer_norm = hidden_act
        self.net = FlatMLP(in_dim, out_dim, hidden_sizes, self.device,
                           hidden_act=hidden_act, output_act=output_act,
                           layer_norm=layer_norm)
        _reference_init(self.net, hidden_w_init, hidden_b_init, output_w_init,
                        output_b_init)
        self._learn_std = bool(learn_std)
        self.net.params[0] = math.log(init_std)
        self._min_log_std = None if min_std is None else float(
            torch.Tensor([min_std]).log())
        self._max_log_std = None if max_std is None else float(
            torch.Tensor([max_std]).log())

    # -- std handling ---------------------------------------------------------
    _std_softplus = False  # std_parameterization='softplus' (Gaussian policy only)

    def _std_args(self):
        """(has_min, min, has_max, max) as the kernels take them; bit 1 of
        ``has_min`` says "softplus parameterisation" (``ga_log_std``, common.h)."""
        return (int(self._min_log_std is not None) |
                (2 if self._std_softplus else 0),
                float(self._min_log_std or 0.0),
                int(self._max_log_std is not None),
                float(self._max_log_std or 0.0))

    def log_std_of(self, raw):
        """``(log std, d log std / d parameter)`` for a raw parameter value:
        clamp, then ``exp`` (identity on the log) or ``softplus`` -- std =
        log(1 + exp(exp(p))) -- in fp32 like
        ``GaussianMLPBaseModule.forward`` (gaussian_mlp_module.py:165-181)."""
        p, chain = np.float32(raw), np.float32(1.0)
        if self._min_log_std is not None and p < self._min_log_std:
            p, chain = np.float32(self._min_log_std), np.float32(0.0)
        if self._max_log_std is not None and p > self._max_log_std:
            p, chain = np.float32(self._max_log_std), np.float32(0.0)
        if self._std_softplus:
            with np.errstate(over='ignore'):
                e = np.exp(p, dtype=np.float32)
                sp = np.log(np.float32(1.0) + np.exp(e, dtype=np.float32),
                            dtype=np.float32)
                chain = chain * e / ((np.float32(1.0) + np.exp(-e, dtype=np.float32))
                                     * sp)
                p = np.log(sp, dtype=np.float32)
        return float(p), float(chain)

    def clamped_log_std(self):
        """Host value of the policy's log std (clamped parameter through the std
        parameterisation): one tiny D2H copy."""
        return self.log_std_of(float(self.net.params[0].item()))[0]

    # -- torch.nn.Module-like surface -----------------------------------------
    def _std_key(self):
        return '_init_std' if self._learn_std else 'init_std'

    def state_dict(self):
        sd = OrderedDict()
        for key, view in self.net.named_views():
            if key == '_init_std':
                sd[self._prefix + self._std_key()] = view.detach().cpu().clone()
                if self._min_log_std is not None:
                    sd[self._prefix + 'min_std_param'] = torch.tensor(
                        [self._min_log_std])
                if self._max_log_std is not None:
                    sd[self._prefix + 'max_std_param'] = torch.tensor(
                        [self._max_log_std])
            else:
                sd[self._prefix + key] = view.detach().cpu().clone()
        return sd

    def load_state_dict(self, sd):
        for key, view in self.net.named_views():
            name = self._prefix + (self._std_key()
                                   if key == '_init_std' else key)
            src = torch.as_tensor(np.asarray(sd[name]) if not torch.is_tensor(
                sd[name]) else sd[name])
            view.copy_(src.to(self.device, torch.float32).reshape(view.shape))
        for attr, key in (('_min_log_std', 'min_std_param'),
                          ('_max_log_std', 'max_std_param')):
            if self._prefix + key in sd:
                setattr(self, attr, float(sd[self._prefix + key]))

    def parameters(self):
        """Views of the trainable tensors, in the reference's order."""
        views = self.net.named_views()
        if not self._learn_std:
            views = views[1:]
        return [v for _, v in views]

    def named_parameters(self):
        out = []
        for key, view in self.net.named_views():
            if key == '_init_std' and not self._learn_std:
                continue
            out.append((self._prefix + key, view))
        return out

    def get_param_values(self):
        return self.state_dict()

    def set_param_values(self, state_dict):
        self.load_state_dict(state_dict)

    # The rest of the ``nn.Module`` surface launchers and garage's own code touch
    # (``torch/policies/policy.py:9-79``, ``trainer.py``, ``set_gpu_mode`` helpers):
    # there are no sub-modules, no autograd graph and no train / eval distinction
    # (no dropout, no batch norm), so these are bookkeeping only.
    training = True

    def train(self, mode=True):
        self.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)

    def to(self, device=None, *args, **kwargs):
        """Parameters live in HBM from construction; moving to another HIP device
        copies the flat buffers, a CPU target is refused (no CPU fallback)."""
        if device is None or isinstance(device, torch.dtype):
            return self
        device = torch.device(device)
        if device.type != 'cuda':
            raise RuntimeError(
                'garage_amd modules live on the GPU; there is no CPU fallback')
        if device.index is not None and device != self.device:
            net = FlatMLP(self.net.in_dim, self.net.out_dim,
                          self.net.hidden_sizes, device,
                          hidden_act=self.net.hidden_act,
                          output_act=self.net.output_act,
                          layer_norm=self.net.layer_norm)
            for k in ('params', 'grads', 'exp_avg', 'exp_avg_sq'):
                getattr(net, k).copy_(getattr(self.net, k))
            if getattr(self.net, 'opt_state3', None) is not None:
                net.opt_state3 = self.net.opt_state3.to(device)
            net.adam_steps = self.net.adam_steps
            self.net, self.device = net, device
        return self

    def cuda(self, device=None):
        return self.to(torch.device('cuda', torch.cuda.current_device()
                                    if device is None else device))

    def zero_grad(self, set_to_none=False):
        del set_to_none
        self.net.grads.zero_()

    def modules(self):
        return iter([self])

    def children(self):
        return iter([])

    def named_modules(self, memo=None, prefix=''):
        del memo
        return iter([(prefix, self)])

    def buffers(self):
        """``min_std_param`` / ``max_std_param`` are buffers in the reference
        (``gaussian_mlp_module.py:131-152``)."""
        out = []
        for v in (self._min_log_std, self._max_log_std):
            if v is not None:
                out.append(torch.tensor([v], device=self.device))
        return out

    def apply(self, fn):
        fn(self)
        return self

    def requires_grad_(self, requires_grad=True):
        del requires_grad  # gradients exist only inside the update kernels
        return self

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)

    def copy_params_from(self, other):
        """Device-to-device parameter copy (``_old_policy`` sync)."""
        self.net.params.copy_(other.net.params)

    def _as_device_obs(self, observations):
        """Accept numpy / torch / list observations, return padded (B, ldo)."""
        if torch.is_tensor(observations) and observations.is_cuda and \
                observations.dim() == 2 and \
                observations.shape[1] == round4(self.net.in_dim) and \
                observations.dtype == torch.float32:
            return observations
        if isinstance(observations, (list, tuple)):
            observations = np.stack([np.asarray(o) for o in observations])
        if torch.is_tensor(observations):
            observations = observations.detach().cpu().numpy()
        space = getattr(getattr(self, '_env_spec', None), 'observation_space',
                        None)
        obs = np.asarray(observations)
        if space is not None and is_discrete(space) and (
                obs.ndim == 1 or obs.shape[-1] != space.flat_dim):
            # state indices -> one-hot rows (observation_space.flatten_n,
            # torch/policies/stochastic_policy.py:70-74)
            ids = obs.astype(np.int64).reshape(-1)
            flat = np.zeros((ids.shape[0], space.flat_dim), dtype=np.float32)
            flat[np.arange(ids.shape[0]), ids] = 1.0
            return pad_rows(flat)
        flat = np.asarray(observations, dtype=np.float32)
        if flat.ndim == 1 and flat.shape[0] == self.net.in_dim:
            flat = flat[None]  # a single observation
        flat = flat.reshape(flat.shape[0], -1)
        if flat.shape[1] != self.net.in_dim:
            raise ValueError(
                'observations of width {} given to a network with {} inputs'.format(
                    flat.shape[1], self.net.in_dim))
        return pad_rows(flat)

    # -- pickling (Trainer snapshots cloudpickle the algo, trainer.py:263-293)
    def __getstate__(self):
        state = {k: v for k, v in self.__dict__.items()
                 if k not in ('net', 'device')}
        state['_hidden_sizes'] = self.net.hidden_sizes
        state['_hidden_act'] = self.net.hidden_act
        state['_output_act'] = self.net.output_act
        state['_layer_norm'] = self.net.layer_norm
        state['_dims'] = (self.net.in_dim, self.net.out_dim)
        for k in ('params', 'exp_avg', 'exp_avg_sq'):
            state['_net_' + k] = getattr(self.net, k).cpu().numpy()
        state['_net_steps'] = self.net.adam_steps
        if getattr(self.net, 'opt_state3', None) is not None:
            # third state buffer of a non-default optimizer (amsgrad / centred)
            state['_net_opt_state3'] = self.net.opt_state3.cpu().numpy()
        return state

    def __setstate__(self, state):
        hidden = state.pop('_hidden_sizes')
        act = state.pop('_hidden_act', 'tanh')
        out_act = state.pop('_output_act', 'none')
        layer_norm = state.pop('_layer_norm', False)
        in_dim, out_dim = state.pop('_dims')
        bufs = {k: state.pop('_net_' + k)
                for k in ('params', 'exp_avg', 'exp_avg_sq')}
        steps = state.pop('_net_steps')
        state3 = state.pop('_net_opt_state3', None)
        self.__dict__.update(state)
        self.device = require_gpu()
        self.net = FlatMLP(in_dim, out_dim, hidden, self.device, hidden_act=act,
                           output_act=out_act, layer_norm=layer_norm)
        for k, v in bufs.items():
            getattr(self.net, k).copy_(torch.from_numpy(v))
        if state3 is not None:
            self.net.opt_state3 = torch.from_numpy(state3).to(self.device)
        self.net.adam_steps = steps


class GaussianMLPPolicy(_GaussianMLP):
    """``garage.torch.policies.GaussianMLPPolicy`` on HIP kernels."""

    def __init__(self,
                 env_spec,
                 hidden_sizes=(32, 32),
                 hidden_nonlinearity=torch.tanh,
                 hidden_w_init=nn.init.xavier_uniform_,
                 hidden_b_init=nn.init.zeros_,
                 output_nonlinearity=None,
                 output_w_init=nn.init.xavier_uniform_,
                 output_b_init=nn.init.zeros_,
                 learn_std=True,
                 init_std=1.0,
                 min_std=1e-6,
                 max_std=None,
                 std_parameterization='exp',
                 layer_normalization=False,
                 name='GaussianMLPPolicy',
                 device=None):
        act = _check_supported(hidden_nonlinearity, output_nonlinearity,
                               std_parameterization, layer_normalization)
        if is_discrete(env_spec.action_space):
            raise ValueError('GaussianMLPPolicy needs a continuous action '
                             'space')
        self._env_spec = env_spec
        self._name = name
        self._obs_dim = env_spec.observation_space.flat_dim
        self._action_dim = env_spec.action_space.flat_dim
        self._build(self._obs_dim, self._action_dim, hidden_sizes,
                    hidden_w_init, hidden_b_init, output_w_init, output_b_init,
                    learn_std, init_std, min_std, max_std, device,
                    hidden_act=act)
        self._std_softplus = std_parameterization == 'softplus'
        # derived from torch's seed without consuming the global stream, so the
        # objects constructed after this one still match the reference's init
        self._sample_seed = int(torch.initial_seed() & 0x7FFFFFFF)
        self._sample_calls = 0

    kind = 'gaussian'

    @property
    def name(self):
        return self._name

    @property
    def env_spec(self):
        return self._env_spec

    @property
    def observation_space(self):
        return self._env_spec.observation_space

    @property
    def action_space(self):
        return self._env_spec.action_space

    def reset(self, do_resets=None):
        """Stateless policy: nothing to reset (``np/policies/policy.py:40``)."""

    def mean(self, obs_dev, M=None, row_idx=None):
        """Policy means ``(M, round4(A))`` for padded device observations."""
        M = obs_dev.shape[0] if M is None else M
        return self.net.forward(obs_dev, M, row_idx=row_idx)

    def forward(self, observations):
        """``(Independent(Normal(mean, std), 1), dict(mean=, log_std=))``.

        ``gaussian_mlp_policy.py:89-102``; the mean comes from the HIP MLP, the
        distribution object is built on the device for callers that want
        ``log_prob`` / ``entropy`` (diagnostics, user code).
        """
        lead = None
        if torch.is_tensor(observations) and observations.dim() > 2:
            lead = observations.shape[:-1]
            observations = observations.reshape(-1, observations.shape[-1])
        elif np.ndim(observations) == 1 and not is_discrete(
                self._env_spec.observation_space):
            lead = ()  # one observation: batch shape () like the reference module
        obs = self._as_device_obs(observations)
        mean = self.mean(obs)[:, :self._action_dim].clone()
        log_std = torch.full_like(mean, self.clamped_log_std())
        if lead is not None:
            mean = mean.reshape(tuple(lead) + (self._action_dim, ))
            log_std = log_std.reshape(mean.shape)
        dist = Independent(Normal(mean, log_std.exp()), 1)
        return dist, dict(mean=mean, log_std=log_std)

    __call__ = forward

    def get_actions(self, observations):
        """``stochastic_policy.py:46-89``: numpy in, numpy out."""
        obs = self._as_device_obs(observations)
        n = obs.shape[0]
        mean = self.mean(obs)
        s = self.clamped_log_std()
        self._sample_calls += 1
        gen = torch.Generator(device=self.device)
        gen.manual_seed(self._sample_seed + self._sample_calls)
        noise = torch.randn(n, self._action_dim, device=self.device,
                            generator=gen)
        m = mean[:, :self._action_dim]
        actions = m + math.exp(s) * noise
        return actions.cpu().numpy(), dict(
            mean=m.cpu().numpy(),
            log_std=np.full((n, self._action_dim), s, dtype=np.float32))

    def get_action(self, observation):
        a, info = self.get_actions(np.asarray(observation)[None])
        return a[0], {k: v[0] for k, v in info.items()}


class CategoricalMLPPolicy(_GaussianMLP):
    """Categorical policy over a discrete action space (BASELINE.json configs 1-2).

    The reference snapshot has **no** torch ``CategoricalMLPPolicy`` (SURVEY.md
    Q15/Q24); this class follows the conventions of its torch categorical (CNN)
    policies and of the TF ``CategoricalMLPPolicy``: tanh MLP with
    ``hidden_sizes=(32, 32)``, xavier-uniform weights, zero biases, a softmax
    output that is then passed as ``logits=`` to ``Categorical``
    (``torch/policies/categorical_cnn_policy.py:138-139``;
    ``double_softmax=True``, the default) -- set ``double_softmax=False`` to
    treat the MLP output as logits.  Actions come back as int64 ``(n,)`` and are
    cast to float for the update exactly like ``torch.Tensor(eps.actions)``
    (``vpg.py:163``).  Parameter names reuse the Gaussian module's scheme
    (``_module._mean_module._layers.{i}.linear.{weight,bias}`` ...).

    Parity: pinned against the reference's ``CategoricalCNNPolicy`` configured as
    an MLP (one 1 x 1 convolution over a ``(O, 1, 1)`` observation is a dense
    layer) through two real ``PPO`` / ``VPG._train_once`` iterations per case
    (``tests/golden/train_once_categorical.npz``,
    ``tests/test_ppo_gpu.py::test_categorical_train_once_matches_real_reference``).
    """

    kind = 'categorical'

    def __init__(self,
                 env_spec,
                 hidden_sizes=(32, 32),
                 hidden_nonlinearity=torch.tanh,
                 hidden_w_init=nn.init.xavier_uniform_,
                 hidden_b_init=nn.init.zeros_,
                 output_w_init=nn.init.xavier_uniform_,
                 output_b_init=nn.init.zeros_,
                 layer_normalization=False,
                 double_softmax=True,
                 name='CategoricalMLPPolicy',
                 device=None):
        act = _check_supported(hidden_nonlinearity, None, 'exp',
                               layer_normalization)
        if not is_discrete(env_spec.action_space):
            raise ValueError('CategoricalMLPPolicy only works '
                             'with akro.Discrete action space.')
        self._env_spec = env_spec
        self._name = name
        self.double_softmax = bool(double_softmax)
        self._obs_dim = env_spec.observation_space.flat_dim
        self._action_dim = env_spec.action_space.n
        self._build(self._obs_dim, self._action_dim, hidden_sizes,
                    hidden_w_init, hidden_b_init, output_w_init, output_b_init,
                    False, 1.0, None, None, device, hidden_act=act)
        self._sample_seed = int(torch.initial_seed() & 0x7FFFFFFF)
        self._sample_calls = 0

    @property
    def name(self):
        return self._name

    @property
    def env_spec(self):
        return self._env_spec

    def reset(self, do_resets=None):
        """Stateless."""

    def state_dict(self):
        sd = super().state_dict()
        sd.pop(self._prefix + 'init_std', None)  # no std in this head
        return sd

    def load_state_dict(self, sd):
        sd = dict(sd)
        sd.setdefault(self._prefix + 'init_std', torch.zeros(1))
        super().load_state_dict(sd)

    def _probs(self, scores):
        p = torch.softmax(scores, dim=-1)
        return torch.softmax(p, dim=-1) if self.double_softmax else p

    def forward(self, observations):
        """``(Categorical, {})`` built from the HIP MLP's scores."""
        obs = self._as_device_obs(observations)
        scores = self.net.forward(obs, obs.shape[0])[:, :self._action_dim]
        probs = self._probs(scores.clone())
        return torch.distributions.Categorical(probs=probs), {}

    __call__ = forward

    def get_actions(self, observations):
        obs = self._as_device_obs(observations)
        scores = self.net.forward(obs, obs.shape[0])[:, :self._action_dim]
        probs = self._probs(scores.clone())
        self._sample_calls += 1
        gen = torch.Generator(device=self.device)
        gen.manual_seed(self._sample_seed + self._sample_calls)
        a = torch.multinomial(probs, 1, generator=gen)[:, 0]
        return a.cpu().numpy(), dict(prob=probs.cpu().numpy())

    def get_action(self, observation):
        a, info = self.get_actions(np.asarray(observation)[None])
        return a[0], {k: v[0] for k, v in info.items()}


class GaussianMLPValueFunction(_GaussianMLP):
    """``garage.torch.value_functions.GaussianMLPValueFunction`` on HIP kernels."""

    _prefix = 'module.'

    def __init__(self,
                 env_spec,
                 hidden_sizes=(32, 32),
                 hidden_nonlinearity=torch.tanh,
                 hidden_w_init=nn.init.xavier_uniform_,
                 hidden_b_init=nn.init.zeros_,
                 output_nonlinearity=None,
                 output_w_init=nn.init.xavier_uniform_,
                 output_b_init=nn.init.zeros_,
                 learn_std=True,
                 init_std=1.0,
                 layer_normalization=False,
                 name='GaussianMLPValueFunction',
                 device=None):
        act = _check_supported(hidden_nonlinearity, output_nonlinearity, 'exp',
                               layer_normalization)
        self._env_spec = env_spec
        self.name = name
        self._build(env_spec.observation_space.flat_dim, 1, hidden_sizes,
                    hidden_w_init, hidden_b_init, output_w_init, output_b_init,
                    learn_std, init_std, None, None, device, hidden_act=act)

    def values(self, obs_dev, M=None, row_idx=None):
        """``(M, 4)`` device tensor, the value in column 0."""
        M = obs_dev.shape[0] if M is None else M
        return self.net.forward(obs_dev, M, row_idx=row_idx)

    def forward(self, obs):
        """``gaussian_mlp_value_function.py:100-112``: ``(..., O) -> (...)``."""
        lead = None
        if torch.is_tensor(obs) and obs.dim() > 2:
            lead = obs.shape[:-1]
            obs = obs.reshape(-1, obs.shape[-1])
        elif np.ndim(obs) == 1:
            lead = ()  # one observation -> a scalar
        v = self.values(self._as_device_obs(obs))[:, 0].clone()
        return v if lead is None else v.reshape(tuple(lead))

    __call__ = forward

    def compute_loss(self, obs, returns):
        """Gaussian NLL of ``returns`` (``gaussian_mlp_value_function.py:81-98``)."""
        x = self._as_device_obs(obs)
        M = x.shape[0]
        v = self.values(x)
        ret = torch.as_tensor(returns).to(self.device,
                                          torch.float32).reshape(-1)
        out = torch.zeros(1, dtype=torch.float32, device=self.device)
        call('ga_gaussian_nll_loss_f32', dptr(v), v.stride(0), dptr(ret), None,
             dptr(self.net.log_std), M, None, dptr(out), None, 0, 0,
             dptr(reduction_workspace(self.device)), stream_ptr())
        return out[0]


__all__ = ['GaussianMLPPolicy', 'GaussianMLPValueFunction']
