"""Oracle: TRPO policy step on the CPU (torch autograd, double backward).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Restates
  * ``torch/algos/trpo.py:93-144``  unclipped surrogate, one full-batch policy
    step per iteration through the constrained optimizer
  * ``torch/optimizers/conjugate_gradient_optimizer.py:18-66``  Hessian-vector
    product of the KL constraint by double backward (Pearlmutter) + reg * v
  * ``...:69-104``   conjugate gradient (10 iterations, residual_tol 1e-10)
  * ``...:146-186``  step: direction, NaN scrub, step size
    sqrt(2 delta / (s^T A s + 1e-8))
  * ``...:236-277``  backtracking line search over ratio**k, k < max_backtracks;
    accept when the loss improves and the constraint holds, else restore
Pinned by ``tests/golden/trpo_train_once.npz`` (the real classes, recorded CG
direction / descent step / post-step parameters).
"""
import numpy as np
import torch

from oracle import networks as nets
from oracle.ppo import OraclePPO


def build_hessian_vector_product(func, params, reg_coeff=1e-5):
    """``conjugate_gradient_optimizer.py:18-66``."""
    shapes = [p.shape or torch.Size([1]) for p in params]
    f = func()
    f_grads = torch.autograd.grad(f, params, create_graph=True)

    def _eval(vector):
        parts, off = [], 0
        for shp in shapes:
            n = int(np.prod(shp))
            parts.append(vector[off:off + n].reshape(shp))
            off += n
        gvp = torch.sum(torch.stack(
            [torch.sum(g * x) for g, x in zip(f_grads, parts)]))
        hvp = list(torch.autograd.grad(gvp, params, retain_graph=True,
                                       allow_unused=True))
        for i, (hx, p) in enumerate(zip(hvp, params)):
            if hx is None:
                hvp[i] = torch.zeros_like(p)
        flat = torch.cat([h.reshape(-1) for h in hvp])
        return flat + reg_coeff * vector

    return _eval


def conjugate_gradient(f_Ax, b, cg_iters, residual_tol=1e-10):
    """``conjugate_gradient_optimizer.py:69-104`` (Demmel p. 312)."""
    p = b.clone()
    r = b.clone()
    x = torch.zeros_like(b)
    rdotr = torch.dot(r, r)
    for _ in range(cg_iters):
        z = f_Ax(p)
        v = rdotr / torch.dot(p, z)
        x += v * p
        r -= v * z
        newrdotr = torch.dot(r, r)
        mu = newrdotr / rdotr
        p = r + mu * p
        rdotr = newrdotr
        if rdotr < residual_tol:
            break
    return x


class OracleCGOptimizer:
    """``ConjugateGradientOptimizer`` over an explicit parameter list."""

    def __init__(self, params, max_constraint_value, cg_iters=10,
                 max_backtracks=15, backtrack_ratio=0.8, hvp_reg_coeff=1e-5,
                 accept_violation=False):
        self.params = list(params)
        self.max_constraint_value = max_constraint_value
        self.cg_iters = cg_iters
        self.max_backtracks = max_backtracks
        self.backtrack_ratio = backtrack_ratio
        self.hvp_reg_coeff = hvp_reg_coeff
        self.accept_violation = accept_violation
        self.trace = {}
        # test probes (per-iterate pins against the reference's recorded CG run):
        # vectors to multiply by the constraint Hessian, and a descent step whose
        # first `probe_candidates` backtracking candidates are evaluated, both at
        # the parameters the step starts from
        self.probe_vectors = None
        self.probe_descent = None
        self.probe_candidates = 0

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def step(self, f_loss, f_constraint):
        params = [p for p in self.params if p.grad is not None]
        flat_loss_grads = torch.cat([p.grad.reshape(-1) for p in params])
        f_Ax = build_hessian_vector_product(f_constraint, params,
                                            self.hvp_reg_coeff)
        step_dir = conjugate_gradient(f_Ax, flat_loss_grads, self.cg_iters)
        step_dir[step_dir.ne(step_dir)] = 0.
        step_size = np.sqrt(2.0 * self.max_constraint_value *
                            (1. / (torch.dot(step_dir, f_Ax(step_dir)) + 1e-8)))
        if np.isnan(step_size):
            step_size = 1.
        descent_step = step_size * step_dir
        self.trace = dict(grad=flat_loss_grads.detach().numpy().copy(),
                          step_dir=step_dir.detach().numpy().copy(),
                          descent_step=descent_step.detach().numpy().copy())
        if self.probe_vectors is not None:
            self.trace['probe_Ax'] = np.stack([
                f_Ax(torch.as_tensor(v)).detach().numpy().copy()
                for v in self.probe_vectors])
        if self.probe_descent is not None:
            self.trace['probe_ls'] = self._probe_line_search(
                params, torch.as_tensor(self.probe_descent), f_loss, f_constraint)
        self._backtracking_line_search(params, descent_step, f_loss,
                                       f_constraint)

    def _probe_line_search(self, params, descent_step, f_loss, f_constraint):
        """(loss_before, [(loss, constraint) of candidate k]) for a GIVEN descent
        step; parameters are restored."""
        prev = [p.detach().clone() for p in params]
        out = [float(f_loss())]
        off = 0
        steps = []
        for p in params:
            n = p.numel()
            steps.append(descent_step[off:off + n].reshape(p.shape))
            off += n
        for k in range(self.probe_candidates):
            ratio = self.backtrack_ratio**k
            with torch.no_grad():
                for step, pv, p in zip(steps, prev, params):
                    p.copy_(pv - ratio * step)
            out.append((float(f_loss()), float(f_constraint())))
        with torch.no_grad():
            for pv, p in zip(prev, params):
                p.copy_(pv)
        return out

    def _backtracking_line_search(self, params, descent_step, f_loss,
                                  f_constraint):
        prev = [p.detach().clone() for p in params]
        ratios = self.backtrack_ratio**np.arange(self.max_backtracks)
        loss_before = f_loss()
        steps, off = [], 0
        for p in params:
            n = p.numel()
            steps.append(descent_step[off:off + n].reshape(p.shape))
            off += n
        accepted = -1
        for k, ratio in enumerate(ratios):
            with torch.no_grad():
                for step, pv, p in zip(steps, prev, params):
                    p.copy_(pv - ratio * step)
            loss = f_loss()
            constraint_val = f_constraint()
            if (loss < loss_before
                    and constraint_val <= self.max_constraint_value):
                accepted = k
                break
        if ((torch.isnan(loss) or torch.isnan(constraint_val)
             or loss >= loss_before
             or constraint_val >= self.max_constraint_value)
                and not self.accept_violation):
            accepted = -1
            with torch.no_grad():
                for pv, p in zip(prev, params):
                    p.copy_(pv)
        self.trace['accepted'] = accepted


class OracleTRPO(OraclePPO):
    """CPU TRPO with garage semantics (Gaussian policy)."""

    def __init__(self, policy_params, value_params, *, max_episode_length,
                 max_constraint_value=0.01, cg_iters=10, max_backtracks=15,
                 backtrack_ratio=0.8, hvp_reg_coeff=1e-5,
                 accept_violation=False, gae_lambda=0.98, **kw):
        super().__init__(policy_params, value_params,
                         max_episode_length=max_episode_length, algo='trpo',
                         gae_lambda=gae_lambda, **kw)
        self.cg = OracleCGOptimizer(
            [self.policy[k] for k in nets.trainable_keys(self.policy)],
            max_constraint_value, cg_iters, max_backtracks, backtrack_ratio,
            hvp_reg_coeff, accept_violation)

    def _objective(self, adv, obs, actions):
        """``trpo.py:93-119``: likelihood ratio times advantage, no clip."""
        new_ll = self._log_prob(self._dist(self.policy, obs), actions)
        with torch.no_grad():
            old_ll = self._log_prob(self._dist(self.old_policy, obs), actions)
        return (new_ll - old_ll).exp() * adv

    def _update_policy(self, obs_flat, actions_flat, adv_flat, used,
                       stream=None):
        """``trpo.py:121-144`` (policy OptimizerWrapper: one full batch)."""
        self.cg.zero_grad()
        loss = self._policy_loss(obs_flat, actions_flat, adv_flat)
        loss.backward()

        def f_loss():
            with torch.no_grad():
                return self._policy_loss(obs_flat, actions_flat, adv_flat)

        def f_constraint():
            return self._kl(obs_flat)

        self.cg.step(f_loss=f_loss, f_constraint=f_constraint)
