"""Device-side building blocks: flat MLP parameter buffers and kernel wrappers.

Everything here launches the hand-written HIP kernels through the C ABI
(``garage_amd._lib``); torch is used for device memory and streams only.
"""
import ctypes as C
import math

import numpy as np
import torch

from garage_amd import _lib
from garage_amd._lib import call, dptr, stream_ptr


def round4(v):
    return (int(v) + 3) // 4 * 4


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError(
            'garage_amd needs an MI355X (HIP device); there is no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


_WS = {}


def reduction_workspace(device, tag=0):
    """Per-device (and per concurrent stream: ``tag``) fp64 reduction scratch."""
    key = (device.type, device.index, tag)
    if key not in _WS:
        n = int(_lib.load().ga_reduction_workspace_doubles())
        # zeroed once: the last slot is the ticket of the single-launch
        # reductions, which every launch leaves at 0
        _WS[key] = torch.zeros(n, dtype=torch.float64, device=device)
    return _WS[key]


class FlatMLP:
    """One MLP (tanh, relu or linear hidden layers, + a scalar log-std slot) as a
    single padded fp32 buffer.

    Layout (floats): ``[log_std, 0, 0, 0]`` then, per linear layer,
    ``W[out][round4(in)]`` and ``b[round4(out)]``.  Gradients, Adam moments and
    the split-K gradient slabs use the same layout, so the optimiser is one
    elementwise kernel over the whole buffer.  Hidden sizes and parameter names
    follow ``GaussianMLPModule`` (torch/modules/gaussian_mlp_module.py:195-305).
    """

    # ga_mlp_desc.hidden_act
    HIDDEN_ACTS = {'tanh': 0, 'relu': 1, 'none': 2, 'sigmoid': 3, 'elu': 4,
                   'leaky_relu': 5, 'softplus': 6}
    # ga_mlp_desc.output_act
    OUTPUT_ACTS = {'none': 0, 'tanh': 1, 'relu': 2, 'sigmoid': 3, 'elu': 4,
                   'leaky_relu': 5, 'softplus': 6}

    def __init__(self, in_dim, out_dim, hidden_sizes, device, hidden_act='tanh',
                 output_act='none', layer_norm=False):
        self.hidden_act = hidden_act
        self.output_act = output_act
        self.layer_norm = bool(layer_norm)
        self.in_dim, self.out_dim = int(in_dim), int(out_dim)
        self.hidden_sizes = tuple(int(h) for h in hidden_sizes)
        dims = (self.in_dim, ) + self.hidden_sizes + (self.out_dim, )
        if len(dims) - 1 > 8:
            raise ValueError('at most 7 hidden layers are supported')
        self.dims = dims
        self.device = device
        off = 4
        self.w_off, self.b_off = [], []
        for l in range(len(dims) - 1):
            self.w_off.append(off)
            off += dims[l + 1] * round4(dims[l])
            self.b_off.append(off)
            off += round4(dims[l + 1])
        # layer_normalization: gamma_l, beta_l of the LayerNorm in front of hidden
        # layer l (multi_headed_mlp_module.py:77-81), behind everything else so that
        # the other offsets do not depend on the option
        self.ln_off = []
        if self.layer_norm:
            for l in range(len(dims) - 2):
                self.ln_off.append(off)
                off += 2 * round4(dims[l])
        self.n_flat = off
        self.act_off, aoff = [], 0
        for l in range(len(dims) - 2):
            self.act_off.append(aoff)
            aoff += round4(dims[l + 1])
        # ... and per row: the normalised input of every hidden layer, (mean, rstd)
        self.lnx_off, self.lns_off = [], []
        if self.layer_norm:
            for l in range(len(dims) - 2):
                self.lnx_off.append(aoff)
                aoff += round4(dims[l])
            for l in range(len(dims) - 2):
                self.lns_off.append(aoff)
                aoff += 4
        self.act_width = aoff  # floats of hidden activations per row
        self.ld_out = round4(self.out_dim)
        self.params = torch.zeros(off, dtype=torch.float32, device=device)
        self.grads = torch.zeros_like(self.params)
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        self.adam_steps = 0
        d = _lib.MlpDesc()
        d.hidden_act = self.HIDDEN_ACTS[hidden_act]
        d.output_act = self.OUTPUT_ACTS[output_act]
        d.layer_norm = int(self.layer_norm)
        for l, o in enumerate(self.ln_off):
            d.ln_off[l] = o
            n = round4(dims[l])
            self.params[o:o + dims[l]] = 1.0  # nn.LayerNorm: weight 1, bias 0
        d.n_layers = len(dims) - 1
        for i, v in enumerate(dims):
            d.dims[i] = v
        for l in range(len(dims) - 1):
            d.w_off[l] = self.w_off[l]
            d.b_off[l] = self.b_off[l]
        # activation offsets are per row *block*: layer l's block starts at
        # act_off[l] * M_capacity; fixed up in _workspace().
        self._desc = d
        self._cap = 0

    # -- parameter access ---------------------------------------------------
    def weight(self, l):
        rows, cols = self.dims[l + 1], self.dims[l]
        w = self.params[self.w_off[l]:self.w_off[l] + rows * round4(cols)]
        return w.view(rows, round4(cols))[:, :cols]

    def bias(self, l):
        return self.params[self.b_off[l]:self.b_off[l] + self.dims[l + 1]]

    @property
    def log_std(self):
        return self.params[0:1]

    def named_views(self, buf=None):
        """(reference-style key suffix, view) pairs in ``parameters()`` order."""
        buf = self.params if buf is None else buf
        out = [('_init_std', buf[0:1])]
        nl = len(self.dims) - 1
        for l in range(nl):
            rows, cols = self.dims[l + 1], self.dims[l]
            w = buf[self.w_off[l]:self.w_off[l] + rows * round4(cols)].view(
                rows, round4(cols))[:, :cols]
            b = buf[self.b_off[l]:self.b_off[l] + rows]
            if l < nl - 1:
                base = '_mean_module._layers.{}.linear.'.format(l)
                if self.layer_norm:  # parameters() order: the LayerNorm comes first
                    o, n = self.ln_off[l], round4(cols)
                    ln = '_mean_module._layers.{}.layer_normalization.'.format(l)
                    out.append((ln + 'weight', buf[o:o + cols]))
                    out.append((ln + 'bias', buf[o + n:o + n + cols]))
            else:
                base = '_mean_module._output_layers.0.linear.'
            out.append((base + 'weight', w))
            out.append((base + 'bias', b))
        return out

    # -- workspaces -----------------------------------------------------------
    def _workspace(self, M):
        if M > self._cap:
            # Ragged batches (C5: ~2.1 M samples +- 0.3 % from one iteration to the
            # next) would otherwise outgrow the workspaces by a few rows every few
            # iterations, and every regrowth reallocates tens of GB (measured at
            # C5: +250 ms for that iteration, reserved memory 75 -> 127 GB): the
            # first growth past 64 K rows takes 3 % headroom.
            cap = int(M)
            if cap > 65536:
                cap = -(-int(cap * 1.03) // 1024) * 1024
            dev = self.device
            self._acts = torch.empty(max(1, cap * self.act_width),
                                     dtype=torch.float32, device=dev)
            self._dacts = torch.empty_like(self._acts)
            self._out = torch.empty(cap * self.ld_out, dtype=torch.float32,
                                    device=dev)
            self._dout = torch.zeros(cap * self.ld_out, dtype=torch.float32,
                                     device=dev)
            self._splits = int(_lib.load().ga_mlp_backward_splits(
                C.byref(self._desc), cap))
            self._slabs = torch.zeros(self._splits * self.n_flat,
                                      dtype=torch.float32, device=dev)
            for l in range(len(self.dims) - 2):
                self._desc.act_off[l] = self.act_off[l] * cap
            for l in range(len(self.lnx_off)):
                self._desc.lnx_off[l] = self.lnx_off[l] * cap
                self._desc.lns_off[l] = self.lns_off[l] * cap
            self._cap = cap

    def train_partials(self, rows):
        """Scratch of the fused optimizer-step kernels for minibatches of up to
        ``rows`` rows (``ga_update_partials_floats``); ``None`` when this
        network's shapes take the per-layer kernels."""
        n = int(_lib.load().ga_update_partials_floats(C.byref(self._desc),
                                                      int(rows)))
        if n <= 0:
            return None
        if getattr(self, '_partials', None) is None or \
                self._partials.numel() < n:
            self._partials = torch.zeros(n, dtype=torch.float32,
                                         device=self.device)
        return self._partials

    def out_view(self, M):
        return self._out[:M * self.ld_out].view(M, self.ld_out)

    def dout_view(self, M):
        return self._dout[:M * self.ld_out].view(M, self.ld_out)

    # -- kernels ----------------------------------------------------------------
    def forward(self, X, M, row_idx=None, out=None, keep_acts=True):
        """``out[M, ld_out] = MLP(X[row_idx])``; X is ``(rows, ldx)`` padded.

        ``keep_acts=False``: the caller wants the outputs only (no ``backward`` /
        ``jvp`` on this pass) -- networks the library can evaluate in one launch
        then leave the activation workspace untouched."""
        self._workspace(M)
        assert X.dtype == torch.float32 and X.stride(-1) == 1
        ldx = X.stride(0) if X.dim() == 2 else round4(self.in_dim)
        if out is None:
            out = self.out_view(M)
        acts = self._acts
        if not keep_acts and M >= self.EVAL_MIN_ROWS and \
                _lib.load().ga_mlp_forward_eval_supported(C.byref(self._desc)):
            acts = None
        call('ga_mlp_forward_f32', C.byref(self._desc), dptr(self.params),
             dptr(X), ldx, dptr(row_idx), M, dptr(acts), dptr(out),
             out.stride(0), stream_ptr())
        return out

    # below this the per-layer kernels' small-M tiles are as fast
    EVAL_MIN_ROWS = 4096

    def head_fusable(self):
        """Can the last layer be computed inside the loss kernel
        (``ga_head_*_loss_f32``)?"""
        return (len(self.dims) >= 3 and self.output_act == 'none'
                and bool(_lib.load().ga_head_loss_supported(
                    int(self.dims[-2]), int(self.dims[-1]))))

    def forward_hidden(self, X, M, row_idx=None):
        """Hidden layers only; returns ``(H, ldh)`` of the last hidden layer,
        plus the device addresses of the head's weights and bias."""
        self._workspace(M)
        assert X.dtype == torch.float32 and X.stride(-1) == 1
        call('ga_mlp_forward_f32', C.byref(self._desc), dptr(self.params),
             dptr(X), X.stride(0), dptr(row_idx), M, dptr(self._acts), None,
             self.ld_out, stream_ptr())
        L = len(self.dims) - 1
        ldh = round4(self.dims[-2])
        off = self.act_off[L - 2] * self._cap
        H = self._acts[off:off + M * ldh].view(M, ldh)
        return H, self.params[self.w_off[L - 1]:], self.params[self.b_off[L - 1]:]

    def backward(self, X, M, dout, row_idx=None, out=None):
        """Slabs <- gradient of everything but the log-std slot.  ``dout`` is the
        gradient with respect to the network's OUTPUT (``out``: what the last
        ``forward`` wrote unless given); an ``output_nonlinearity`` scales it in
        place by its slope first."""
        ldx = X.stride(0)
        if self.output_act != 'none':
            out = self.out_view(M) if out is None else out
            call('ga_act_slope_mul_f32', dptr(dout), dout.stride(0), dptr(out),
                 out.stride(0), M, self.out_dim, self.OUTPUT_ACTS[self.output_act],
                 stream_ptr())
        splits = min(self._splits,
                     int(_lib.load().ga_mlp_backward_splits(
                         C.byref(self._desc), M)))
        call('ga_mlp_backward_f32', C.byref(self._desc), dptr(self.params),
             dptr(X), ldx, dptr(row_idx), M, dptr(self._acts), dptr(dout),
             dout.stride(0), dptr(self._dacts), dptr(self._slabs), self.n_flat,
             splits, stream_ptr())
        self._used_splits = splits
        return splits

    def jvp(self, X, M, tangent, row_idx=None):
        """``d(output)[M, ld_out]`` for the parameter tangent ``tangent`` (flat
        layout); needs the activations of a ``forward`` at the same rows."""
        self._workspace(M)
        if getattr(self, '_tout', None) is None or \
                self._tout.numel() < M * self.ld_out:
            self._tout = torch.zeros(self._cap * self.ld_out,
                                     dtype=torch.float32, device=self.device)
        tout = self._tout[:M * self.ld_out].view(M, self.ld_out)
        call('ga_mlp_jvp_f32', C.byref(self._desc), dptr(self.params),
             dptr(tangent), dptr(X), X.stride(0), dptr(row_idx), M,
             dptr(self._acts), dptr(self._dacts), dptr(tout), tout.stride(0),
             stream_ptr())
        if self.output_act != 'none':  # tangent of f(z) = f'(z) tz
            out = self.out_view(M)
            call('ga_act_slope_mul_f32', dptr(tout), tout.stride(0), dptr(out),
                 out.stride(0), M, self.out_dim, self.OUTPUT_ACTS[self.output_act],
                 stream_ptr())
        return tout

    def reduce_grads(self, scale=1.0):
        call('ga_reduce_slabs_f32', dptr(self._slabs), self._used_splits,
             self.n_flat, self.n_flat, float(scale), dptr(self.grads),
             stream_ptr())

    def optimizer_step(self, hyper):
        """One step of a torch.optim class other than the default Adam
        (``optimizers._torch_optimizer_hyper``: SGD, RMSprop, Adam / AdamW with
        weight decay or amsgrad) on ``self.grads``.  ``exp_avg`` / ``exp_avg_sq``
        double as the first two state buffers (momentum buffer / square_avg, ...),
        a third one (max_exp_avg_sq, grad_avg) is allocated on first use."""
        import ctypes
        self.adam_steps += 1
        n1, n2, n3 = hyper['needs']
        if n3 and getattr(self, 'opt_state3', None) is None:
            self.opt_state3 = torch.zeros_like(self.params)
        h = (ctypes.c_double * 5)(*[float(v) for v in hyper['h']])
        call('ga_optimizer_step_f32', int(hyper['code']), dptr(self.params),
             dptr(self.grads), dptr(self.exp_avg) if n1 else None,
             dptr(self.exp_avg_sq) if n2 else None,
             dptr(self.opt_state3) if n3 else None, self.n_flat,
             self.adam_steps, h, int(hyper['flags']), stream_ptr())

    def adam_step(self, lr, betas=(0.9, 0.999), eps=1e-8):
        self.adam_steps += 1
        call('ga_adam_step_f32', dptr(self.params), dptr(self.grads),
             dptr(self.exp_avg), dptr(self.exp_avg_sq), self.n_flat,
             self.adam_steps, float(lr), float(betas[0]), float(betas[1]),
             float(eps), stream_ptr())


def pad_rows(x, width=None):
    """Copy a ``(rows, w)`` array/tensor into a zero padded ``(rows, round4(w))``
    fp32 device tensor (the layout every MLP entry point expects)."""
    dev = require_gpu()
    t = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x)
    t = t.to(device=dev, dtype=torch.float32)
    if t.dim() == 1:
        t = t.unsqueeze(1)
    w = t.shape[1]
    ld = round4(w if width is None else width)
    out = torch.zeros(t.shape[0], ld, dtype=torch.float32, device=dev)
    out[:, :w] = t
    return out


def gae_scan(rewards, values, *, discount, gae_lambda, max_episode_length,
             tail=None, offsets=None, max_len=None, v0=0.0, bonus=None,
             bonus_const=0.0, adv=None, ret=None):
    """Returns + GAE advantages (``ga_gae_scan_f32``).

    ``rewards``/``values`` are ``(n_rows, T)`` device tensors (mode 0 with
    ``tail``; mode 1 without), or packed 1-D tensors with ``offsets``.
    """
    if offsets is not None:
        n_rows = offsets.numel() - 1
        T, ld = rewards.numel(), 0  # packed: T carries the total step count
        if max_len is None:
            raise ValueError('max_len is required with offsets')
    else:
        n_rows, T = rewards.shape
        ld = rewards.stride(0)
        assert values.stride(0) == ld
        max_len = T
    if adv is None:
        adv = torch.empty_like(rewards)
    if ret is None:
        ret = torch.empty_like(rewards)
    mode = 0 if tail is not None else 1
    call('ga_gae_scan_f32', dptr(rewards), dptr(values), dptr(bonus),
         dptr(tail), dptr(offsets), n_rows, T, ld, int(max_len), mode,
         int(max_episode_length), float(discount), float(gae_lambda),
         float(v0), float(bonus_const), dptr(adv), dptr(ret), stream_ptr())
    return adv, ret


def center_advantages(adv, *, center=True, positive=False, stats=None,
                      allreduce=None):
    """In-place ``VPG._compute_advantage`` post-processing (vpg.py:371-377).

    ``allreduce(tensor, op)`` (op in {'sum', 'min'}) is called between the
    reduction stages when the batch is sharded over ranks, so every rank
    normalises with the global mean / unbiased variance / minimum.
    """
    dev = adv.device
    ws = reduction_workspace(dev)
    if stats is None:
        stats = torch.zeros(4, dtype=torch.float64, device=dev)
    n = adv.numel()
    s = stream_ptr()
    if center:
        call('ga_stats_f32', dptr(adv), n, 0, dptr(stats), dptr(ws), s)
        if allreduce is not None:
            allreduce(stats[0:2], 'sum')
        call('ga_stats_f32', dptr(adv), n, 1, dptr(stats), dptr(ws), s)
        if allreduce is not None:
            allreduce(stats[2:3], 'sum')
        call('ga_adv_center_f32', dptr(adv), n, dptr(stats), 1e-8, s)
    if positive:
        call('ga_stats_f32', dptr(adv), n, 2, dptr(stats), dptr(ws), s)
        if allreduce is not None:
            allreduce(stats[3:4], 'min')
        call('ga_sub_scalar_f32', dptr(adv), n, dptr(stats[3:4]), s)
    return adv


HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)
