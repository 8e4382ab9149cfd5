"""CONTAINER-ONLY developer script (needs /root/reference; never runs on the GPU
box): times the oracle (``oracle/``: what ``bench.py``'s ``cpu_baseline`` runs, kind
"port") next to the REAL reference (``garage.sampler.LocalSampler(VecWorker)`` +
``garage.torch.algos.PPO``) on identical inputs, so that BASELINE.md can say how
fair a stand-in the port is (within ~10 %).

Same per-env ``SyntheticEnv`` objects (wrapped as ``garage.Environment`` for the
reference), same initial parameters, same hyper-parameters (E = 10, Adam lr 2.5e-4,
gamma 0.99, lambda 0.97, clip 0.2), HalfCheetah shape (obs 17, act 6,
MLP(256, 256)), and the two minibatch settings of SURVEY.md section 8d:
(i) throughput ``mb = S / 32``, (ii) the reference default ``mb = 64``.

    PYTHONDONTWRITEBYTECODE=1 python tools/time_oracle_vs_reference.py [--envs 64] [--T 256]

Prints one JSON object; the update times of both are of ``_train_once`` on the
SAME batch (the reference's own rollout), the rollout times of each one's sampler.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
sys.dont_write_bytecode = True

import numpy as np  # noqa: E402
import torch  # noqa: E402

import _ref_harness as ref  # noqa: E402

ref.install()

import akro  # noqa: E402  (the in-memory stub)
from garage import EnvSpec, Environment, EnvStep, StepType  # noqa: E402
from garage.sampler import LocalSampler, VecWorker  # noqa: E402
from garage.torch.algos import PPO  # noqa: E402
import garage.torch.algos.vpg as vpg_mod  # noqa: E402
import garage._functions as gfun  # noqa: E402
from garage.torch.optimizers import OptimizerWrapper  # noqa: E402
from garage.torch.policies import GaussianMLPPolicy  # noqa: E402
from garage.torch.value_functions import \
    GaussianMLPValueFunction  # noqa: E402

from oracle import batch as ob  # noqa: E402
from oracle import envs as oenvs  # noqa: E402
from oracle import networks as nets  # noqa: E402
from oracle import sampler as osamp  # noqa: E402
from oracle.ppo import OraclePPO  # noqa: E402


class RefEnv(Environment):
    """A real ``garage.Environment`` over the oracle's env definition."""

    def __init__(self, inner, obs_dim, act_dim, P):
        self._inner = inner
        self._obs_space = akro.Box(-np.inf, np.inf, (obs_dim, ))
        self._act_space = akro.Box(-np.inf, np.inf, (act_dim, ))
        self._spec = EnvSpec(self._obs_space, self._act_space,
                             max_episode_length=P)

    action_space = property(lambda self: self._act_space)
    observation_space = property(lambda self: self._obs_space)
    spec = property(lambda self: self._spec)
    render_modes = property(lambda self: [])

    def reset(self):
        return self._inner.reset()

    def step(self, action):
        s = self._inner.step(action)
        return EnvStep(env_spec=self._spec, action=s.action, reward=s.reward,
                       observation=s.observation, env_info=s.env_info,
                       step_type=StepType(int(s.step_type)))

    def render(self, mode):
        pass

    def visualize(self):
        pass

    def close(self):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=64)
    ap.add_argument('--T', type=int, default=256)
    ap.add_argument('--threads', type=int, default=8)
    ap.add_argument('--skip-mb64', action='store_true')
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    O, A, hs, n, T = 17, 6, (256, 256), args.envs, args.T
    S = n * T
    E = 10
    out = dict(envs=n, T=T, samples=S, threads=torch.get_num_threads(),
               logical_cpus=os.cpu_count(), torch=torch.__version__)
    rec = ref.TabularRecorder()
    vpg_mod.tabular = rec
    gfun.tabular = rec

    def make_reference(mb):
        torch.manual_seed(1)
        spec = EnvSpec(akro.Box(-np.inf, np.inf, (O, )),
                       akro.Box(-np.inf, np.inf, (A, )), max_episode_length=T)
        pol = GaussianMLPPolicy(spec, hidden_sizes=hs)
        vf = GaussianMLPValueFunction(spec, hidden_sizes=hs)
        envs = [RefEnv(oenvs.SyntheticEnv(i, O, A, T, seed=1), O, A, T)
                for i in range(n)]
        sampler = LocalSampler(agents=pol, envs=[envs], max_episode_length=T,
                               n_workers=1, worker_class=VecWorker,
                               worker_args=dict(n_envs=n))
        opt = (torch.optim.Adam, dict(lr=2.5e-4))
        algo = PPO(env_spec=spec, policy=pol, value_function=vf,
                   sampler=sampler,
                   policy_optimizer=OptimizerWrapper(
                       opt, pol, max_optimization_epochs=E, minibatch_size=mb),
                   vf_optimizer=OptimizerWrapper(
                       opt, vf, max_optimization_epochs=E, minibatch_size=mb),
                   discount=0.99, gae_lambda=0.97, lr_clip_range=0.2)
        return algo, sampler, pol, vf

    for label, mb in (('mb_S_over_32', S // 32), ('mb_64', 64)):
        if mb == 64 and args.skip_mb64:
            continue
        algo, sampler, pol, vf = make_reference(mb)
        pol0 = {k: v.clone() for k, v in pol.state_dict().items()}
        vf0 = {k: v.clone() for k, v in vf.state_dict().items()}
        # ---- the real reference (one untimed rollout first: the first call
        #      pays for lazy imports and allocator warm-up)
        sampler.obtain_samples(0, S, pol.get_param_values())
        np.random.seed(1)
        torch.manual_seed(2)
        t0 = time.perf_counter()
        eps = sampler.obtain_samples(0, S, pol.get_param_values())
        t1 = time.perf_counter()
        np.random.seed(3)
        algo._train_once(0, eps)
        t2 = time.perf_counter()
        ref_tab = {k: float(v) for k, v in rec.values.items()
                   if k.endswith('LossAfter')}
        # ---- the oracle: its own sampler on the same env definitions, its
        #      update on the SAME batch the reference just trained on
        oracle = OraclePPO(pol0, vf0, max_episode_length=T,
                           max_optimization_epochs=E, minibatch_size=mb,
                           policy_lr=2.5e-4, vf_lr=2.5e-4, discount=0.99,
                           gae_lambda=0.97, lr_clip_range=0.2)

        class Agent:

            def reset(self, do_resets=None):
                pass

            def get_actions(self, obs):
                with torch.no_grad():
                    x = torch.from_numpy(np.asarray(obs, np.float32))
                    dist, info = nets.policy_forward(oracle.policy, x)
                    return dist.sample().numpy(), {
                        k: v.numpy() for k, v in info.items()}

        osampler = osamp.OracleLocalSampler(
            Agent(), [[oenvs.SyntheticEnv(i, O, A, T, seed=1)
                       for i in range(n)]], max_episode_length=T, n_workers=1,
            worker_class=osamp.OracleVecWorker, worker_args=dict(n_envs=n))
        osampler.obtain_samples(0, S, None)  # untimed, as above
        torch.manual_seed(2)
        t3 = time.perf_counter()
        obatch = osampler.obtain_samples(0, S, None)
        t4 = time.perf_counter()
        host = ob.OracleEpisodeBatch(
            observations=eps.observations,
            last_observations=eps.last_observations, actions=eps.actions,
            rewards=eps.rewards,
            step_types=np.asarray([int(s) for s in eps.step_types]),
            lengths=eps.lengths, max_episode_length=T)
        np.random.seed(3)
        t5 = time.perf_counter()
        want = oracle.train_once(host)
        t6 = time.perf_counter()
        # same batch, same permutations, same initial parameters: same results
        wp, _ = oracle.state()
        dmax = max(float(np.abs(np.asarray(wp[k]) -
                                pol.state_dict()[k].numpy()).max())
                   for k in wp)
        steps = int(np.sum(eps.lengths))
        out[label] = dict(
            minibatch=mb, optimizer_steps=2 * E * -(-steps // mb),
            reference=dict(rollout_s=t1 - t0, update_s=t2 - t1,
                           env_steps_per_s=steps / (t2 - t0)),
            oracle=dict(rollout_s=t4 - t3, update_s=t6 - t5,
                        env_steps_per_s=steps / ((t4 - t3) + (t6 - t5)),
                        rollout_steps=int(obatch.lengths.sum())),
            oracle_over_reference=dict(
                rollout=(t4 - t3) / (t1 - t0), update=(t6 - t5) / (t2 - t1),
                iteration=((t4 - t3) + (t6 - t5)) / (t2 - t0)),
            # (NOT a parity figure: the two log-probability formulas differ in
            # rounding and hundreds of Adam steps on small minibatches amplify
            # that; parity is pinned step by step by tests/golden)
            max_abs_policy_param_difference_after_all_steps=dmax,
            oracle_scalars={k: float(want[k]) for k in
                            ('policy/LossAfter', 'vf/LossAfter')},
            reference_scalars=ref_tab)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
