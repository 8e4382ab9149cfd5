#!/usr/bin/env python
"""Headline benchmark: env-steps/s of one PPO iteration (rollout + update).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Without a launcher (``WORLD_SIZE`` unset) ``--gpus N > 1`` starts the N ranks
itself: N fresh child processes with a torchrun-style environment, before this
process has touched the GPU; it relays rank 0's JSON line and exits non-zero
if any rank does.

One "step" = one PPO iteration: a rollout of ``n_envs x T`` synthetic env steps
through ``GpuVecSampler`` followed by one ``PPO._train_once`` (value baselines,
GAE scan, advantage centring, E epochs x 32 minibatches of policy updates, then
the same for the value function, diagnostics, old-policy sync) -- all of it in
the timed region.  Weak scaling: every rank owns ``n_envs`` environments.
Prints ONE JSON line (rank 0) with ``roofline`` (dominant kernel, HIP-event
timed), ``roofline_gae_scan`` and ``cpu_baseline``.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np


def self_launch(n):
    """Run this command as ``n`` ranks (one child process per GPU) and relay
    rank 0's output.  The parent never initialises the GPU."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r),
                   WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen(
            [sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
            env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr))
    rc = 0
    try:
        pending = set(range(n))
        while pending and rc == 0:
            time.sleep(0.2)
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0:
                    print('bench.py: rank {} exited with code {}'.format(
                        r, code), file=sys.stderr)
                    rc = code if code > 0 else 1
    finally:
        for p in procs:  # a failed rank leaves its peers in a collective
            if p.poll() is None:
                if rc == 0:
                    p.wait()
                else:
                    p.terminate()
        out = procs[0].stdout.read().decode()
        for p in procs:
            if p.poll() is None:
                try:
                    p.wait(10)
                except subprocess.TimeoutExpired:
                    p.kill()
    sys.stdout.write(out)
    sys.stdout.flush()
    return rc



def _requested_gpus(argv):
    for i, arg in enumerate(argv):
        if arg == '--gpus' and i + 1 < len(argv):
            return int(argv[i + 1])
        if arg.startswith('--gpus='):
            return int(arg.split('=', 1)[1])
    return 1


if __name__ == '__main__' and 'WORLD_SIZE' not in os.environ \
        and _requested_gpus(sys.argv[1:]) > 1:
    # no launcher around us: become the launcher BEFORE torch (or anything that
    # could initialise the GPU) is imported into this process
    sys.exit(self_launch(_requested_gpus(sys.argv[1:])))

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # BASELINE.json configs[2]: the configuration the metric is quoted on
    'c3': dict(name='HalfCheetah-shape synthetic', obs_dim=17, act_dim=6,
               n_envs=4096, T=256, hidden=(256, 256), min_len=None),
    'c2': dict(name='CartPole-shape synthetic, discrete-2 categorical head',
               obs_dim=4, act_dim=2, n_envs=4096, T=128, hidden=(64, 64),
               min_len=None, discrete=True),
    'c5': dict(name='Humanoid-shape synthetic, ragged', obs_dim=376,
               act_dim=17, n_envs=8192, T=256, hidden=(512, 512, 512),
               min_len=32),
    # SURVEY.md section 8d setting (ii): the reference's default minibatch
    # (ppo.py:65-76: 64 samples) on a reduced batch, 5 120 optimizer steps per
    # iteration as in BASELINE.md section 2's reference measurement
    # BASELINE.json configs[0]'s shape (CartPole: obs 4, 2 discrete actions,
    # MLP(32,32), batch 2048, the reference-default minibatch of 64) on the
    # synthetic env -- the reference itself runs this one on the CPU
    'c1': dict(name='CartPole-shape synthetic, discrete-2 categorical head, '
               'batch 2048, reference-default minibatch 64', obs_dim=4,
               act_dim=2, n_envs=16, T=128, hidden=(32, 32), min_len=None,
               discrete=True, minibatches=32),
    'c3mb64': dict(name='HalfCheetah-shape synthetic, reference-default '
                   'minibatch 64 on a reduced batch', obs_dim=17, act_dim=6,
                   n_envs=64, T=256, hidden=(256, 256), min_len=None,
                   minibatches=256),
}
# BASELINE.json's second metric ("GAE-scan HBM GB/s") at the footprint of the C4
# batch (8 x 4096 envs x T = 256 = 134 MB of r, V, A, G): one step = one scan
SCAN_CONFIGS = {
    'c4scan': dict(name='GAE(lambda) + discounted-return scan, C4 batch '
                   'footprint', n_rows=32768, T=256),
    'c3scan': dict(name='GAE(lambda) + discounted-return scan, C3 batch '
                   'footprint', n_rows=4096, T=256),
}
HYPER = dict(discount=0.99, gae_lambda=0.97, lr_clip_range=0.2, lr=2.5e-4,
             epochs=10, minibatches_per_epoch=32)

CPU_THREADS = int(os.environ.get('GARAGE_AMD_CPU_THREADS', '16'))
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec peak

# garage_amd/csrc/prof.h kinds (kernel names as rocprofv3 prints them)
KIND_NAMES = [
    'gemm_f32_kernel<128,128,2,4,true,true,32,false> (forward)',
    'gemm_f32_kernel<128,128,2,4,true,false,32,false> (data grad)',
    'gemm_f32_kernel<128,128,2,4,false,false,32,false> (weight grad)',
    'gemm_f32_kernel<128,32,4,1,true,true,32,false> (forward, narrow)',
    'gemm_f32_kernel<128,32,4,1,true,false,32,false> (data grad, narrow)',
    'gemm_f32_kernel<128,32,4,1,false,false,32,false> (weight grad, narrow)',
    'gae_scan_rows_kernel / gae_scan_kernel',
    'skinny_fwd_kernel (first-layer forward / head data grad; work = bytes)',
    'skinny_wgrad_kernel (first-layer / head weight grad; work = bytes)',
    'fwd_head_loss_kernel<256,1,8,false> (last hidden layer + head + loss + seed)',
    'dgrad_wgrad0_kernel<256,1,8> (data grad + first-layer weight grad)',
    'narrow_train_kernel<64> (forward + loss + backward of a 2 x 64 net)',
    'mlp_eval_forward_kernel<256,1,8> (whole MLP, outputs only: full-batch passes)',
]
GEMM_KINDS = (0, 1, 2, 3, 4, 5, 9, 10, 11, 12)  # MFMA kernels: work = flops


TRAFFIC_FILE = 'profiles/r03_traffic.json'


def load_traffic():
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (C3)."""
    path = os.path.join(ROOT, TRAFFIC_FILE)
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        return json.load(f).get('kernels', {})


def n_minibatches(cfg):
    return cfg.get('minibatches', HYPER['minibatches_per_epoch'])


def build_engine(cfg, comm, seed=1, algo_name='ppo'):
    from garage_amd.algos import PPO
    from garage_amd.distributed import shard_algo
    from garage_amd.envs import SyntheticVecEnv
    from garage_amd.optimizers import OptimizerWrapper
    from garage_amd.policies import (CategoricalMLPPolicy, GaussianMLPPolicy,
                                     GaussianMLPValueFunction)
    from garage_amd.sampler import GpuVecSampler, GpuVecWorker
    rank = comm.rank if comm is not None else 0
    n, T = cfg['n_envs'], cfg['T']
    torch.manual_seed(seed)
    env = SyntheticVecEnv(n, cfg['obs_dim'], cfg['act_dim'], T,
                          min_len=cfg['min_len'], seed=seed, env_id0=rank * n,
                          discrete=cfg.get('discrete', False))
    policy_cls = (CategoricalMLPPolicy
                  if cfg.get('discrete') else GaussianMLPPolicy)
    pol = policy_cls(env.spec, hidden_sizes=cfg['hidden'])
    vf = GaussianMLPValueFunction(env.spec, hidden_sizes=cfg['hidden'])
    sampler = GpuVecSampler(pol, env, max_episode_length=T, n_workers=1,
                            worker_class=GpuVecWorker, seed=seed + rank,
                            worker_args=dict(n_envs=n))
    S = n * T
    mb = S // n_minibatches(cfg)
    opt = (torch.optim.Adam, dict(lr=HYPER['lr']))
    if algo_name == 'trpo':
        # one conjugate-gradient policy step on the full
        # batch + the same value-function passes (not the headline metric)
        from garage_amd.algos import TRPO
        from garage_amd.optimizers import ConjugateGradientOptimizer
        algo = TRPO(env_spec=env.spec, policy=pol, value_function=vf,
                    sampler=sampler,
                    policy_optimizer=OptimizerWrapper(
                        (ConjugateGradientOptimizer,
                         dict(max_constraint_value=0.01)), pol),
                    vf_optimizer=OptimizerWrapper(
                        opt, vf, max_optimization_epochs=HYPER['epochs'],
                        minibatch_size=mb, permutation='device',
                        seed=2 * seed + 1 + 1000 * rank),
                    discount=HYPER['discount'],
                    gae_lambda=HYPER['gae_lambda'])
        shard_algo(algo, comm)
        return algo, sampler, pol, S
    algo = PPO(env_spec=env.spec, policy=pol, value_function=vf,
               sampler=sampler,
               policy_optimizer=OptimizerWrapper(
                   opt, pol, max_optimization_epochs=HYPER['epochs'],
                   minibatch_size=mb, permutation='device',
                   seed=2 * seed + 1000 * rank),
               vf_optimizer=OptimizerWrapper(
                   opt, vf, max_optimization_epochs=HYPER['epochs'],
                   minibatch_size=mb, permutation='device',
                   seed=2 * seed + 1 + 1000 * rank),
               lr_clip_range=HYPER['lr_clip_range'],
               discount=HYPER['discount'], gae_lambda=HYPER['gae_lambda'])
    shard_algo(algo, comm)
    return algo, sampler, pol, S


def one_iteration(algo, sampler, pol, S, itr):
    eps = sampler.obtain_samples(itr, S, None)
    algo._train_once(itr, eps)


def roofline_pass(algo, sampler, pol, S, itr):
    """One extra, instrumented iteration: every GEMM / scan launch is bracketed
    by HIP events on its stream (garage_amd/csrc/prof.cpp)."""
    from garage_amd import _lib
    lib = _lib.load()
    lib.ga_prof_enable(1)
    one_iteration(algo, sampler, pol, S, itr)
    torch.cuda.synchronize()
    lib.ga_prof_enable(0)
    n_kinds = len(KIND_NAMES)
    out = (C.c_double * (3 * n_kinds))()
    lib.ga_prof_collect(out, n_kinds)
    rows = []
    # layers of at most 64 units run the wide kinds on 64x64 tiles (4 waves)
    small = max(algo.policy.net.hidden_sizes) <= 64
    width = max(algo.policy.net.hidden_sizes)
    for k in range(n_kinds):
        ms, work, cnt = out[3 * k], out[3 * k + 1], out[3 * k + 2]
        name = KIND_NAMES[k]
        if small:
            name = name.replace('<128,128,2,4,', '<64,64,2,2,')
        if width == 64:
            name = name.replace('<256,1,8', '<64,2,2')
        elif width == 128:
            name = name.replace('<256,1,8', '<128,1,4')
        hs = algo.policy.net.hidden_sizes
        in_w = int(algo.policy.net.dims[0])
        if (len(hs) == 2 and in_w <= 32 and hs[0] % 32 == 0
                and hs[0] * ((in_w + 3) // 4 * 4) <= 5120):
            # the first layer is computed inside the kernel (ga_set_fused_first_layer)
            name = name.replace(',false> (last hidden layer',
                                ',true> (first layer + last hidden layer')
            if width == 256 and (in_w + 3) // 4 == 5 and \
                    os.environ.get('GARAGE_AMD_PIPELINED_KLOOP', '1') != '0':
                # the software-pipelined k-loop instantiation (17..20 inputs)
                name = name.replace('<256,1,8,true>', '<256,1,8,true,5>')
                name = name.replace('mlp_eval_forward_kernel<256,1,8>',
                                    'mlp_eval_forward_kernel<256,1,8,5>')
        rows.append(dict(kernel=name, total_ms=ms, work=work,
                         launches=int(cnt)))
    return rows


def cpu_baseline(cfg, n_envs, seed=1):
    """The oracle (CPU restatement of garage's LocalSampler/VecWorker rollout +
    torch-CPU PPO) timed on a bounded sample of the same workload."""
    from oracle import envs as oenvs
    from oracle import networks as nets
    from oracle import sampler as osamp
    from oracle.ppo import OraclePPO
    O, A, T = cfg['obs_dim'], cfg['act_dim'], cfg['T']
    # a one-GPU box owns a 16-CPU share of the host; torch's default (one
    # thread per logical CPU of the whole machine) oversubscribes it badly
    torch.set_num_threads(max(1, min(CPU_THREADS, os.cpu_count() or 1)))
    rng = np.random.RandomState(seed)
    discrete = bool(cfg.get('discrete'))
    polp = nets.init_gaussian_mlp(rng, nets.POLICY_PREFIX, O, A, cfg['hidden'],
                                  min_std=None if discrete else 1e-6)
    vfp = nets.init_gaussian_mlp(rng, nets.VALUE_PREFIX, O, 1, cfg['hidden'])
    S = n_envs * T
    # the sample keeps the GPU run's minibatch size when the config fixes it
    mb = (cfg['n_envs'] * T // cfg['minibatches'] if 'minibatches' in cfg
          else S // HYPER['minibatches_per_epoch'])
    algo = OraclePPO(polp, vfp, max_episode_length=T,
                     max_optimization_epochs=HYPER['epochs'],
                     minibatch_size=mb, policy_lr=HYPER['lr'],
                     vf_lr=HYPER['lr'], discount=HYPER['discount'],
                     gae_lambda=HYPER['gae_lambda'],
                     lr_clip_range=HYPER['lr_clip_range'],
                     policy_kind='categorical' if discrete else 'gaussian')

    class Agent:

        def reset(self, do_resets=None):
            pass

        def get_actions(self, obs):
            with torch.no_grad():
                x = torch.from_numpy(np.asarray(obs, np.float32))
                if discrete:
                    dist = nets.categorical_dist(algo.policy,
                                                 nets.POLICY_PREFIX, x)
                    return dist.sample().numpy(), {}
                dist, info = nets.policy_forward(algo.policy, x)
                return dist.sample().numpy(), {
                    k: v.numpy() for k, v in info.items()
                }

    envs = [oenvs.SyntheticEnv(i, O, A, T, min_len=cfg['min_len'], seed=seed,
                               discrete=discrete)
            for i in range(n_envs)]
    sampler = osamp.OracleLocalSampler(
        Agent(), [envs], max_episode_length=T, n_workers=1,
        worker_class=osamp.OracleVecWorker, worker_args=dict(n_envs=n_envs))
    np.random.seed(seed)
    t0 = time.perf_counter()
    batch = sampler.obtain_samples(0, S, None)
    t1 = time.perf_counter()
    algo.train_once(batch)
    t2 = time.perf_counter()
    steps = int(batch.lengths.sum())
    return dict(value=steps / (t2 - t0), unit='env-steps/s',
                cores=torch.get_num_threads(), kind='port',
                sample='{} envs x T={} ({} steps), same PPO hyper-parameters '
                '(E={}, {} minibatches/epoch); rollout {:.2f} s + update '
                '{:.2f} s; host has {} logical CPUs'.format(
                    n_envs, T, steps, HYPER['epochs'],
                    S // mb, t1 - t0, t2 - t1,
                    os.cpu_count()))


def scan_launches(n, T, steps, sets, warmup=2):
    """``steps`` launches of the returns + GAE scan over ``n`` rows x ``T`` steps,
    each timed with HIP events attached to its dispatch (prof.cpp), rotating
    through ``sets`` distinct (r, V, A, G) buffer sets.  One set is 16 n T bytes;
    with ``sets`` x that well above the 256 MiB Infinity Cache every launch's
    inputs come from HBM and its outputs evict to HBM (``sets == 1``: the same
    buffers back to back, i.e. last-level-cache resident below 256 MiB).
    Returns (GB/s of algorithmic bytes, mean launch us, bytes per launch)."""
    from garage_amd import _lib
    from garage_amd.engine import gae_scan
    lib = _lib.load()
    dev = torch.device('cuda', torch.cuda.current_device())
    g = torch.Generator(device='cpu').manual_seed(0)
    bufs = []
    for _ in range(sets):
        r = torch.randn(n, T, generator=g).to(dev)
        v = torch.randn(n, T, generator=g).to(dev)
        bufs.append((r, v, torch.empty_like(r), torch.empty_like(r)))

    def run(i):
        r, v, adv, ret = bufs[i % sets]
        gae_scan(r, v, discount=HYPER['discount'],
                 gae_lambda=HYPER['gae_lambda'], max_episode_length=T, adv=adv,
                 ret=ret)

    for i in range(max(warmup, sets)):
        run(i)
    torch.cuda.synchronize()
    lib.ga_prof_enable(1)
    for i in range(steps):
        run(i)
    torch.cuda.synchronize()
    lib.ga_prof_enable(0)
    out = (C.c_double * (3 * len(KIND_NAMES)))()
    lib.ga_prof_collect(out, len(KIND_NAMES))
    ms, work, cnt = out[18], out[19], out[20]
    del bufs
    return (work / (ms * 1e-3) / 1e9, ms * 1e3 / max(1.0, cnt),
            work / max(1.0, cnt))


MALL_BYTES = 256 * 2**20  # Infinity Cache (MI355X_MICROARCH.md)


def scan_sets_for_hbm(n, T):
    """Buffer sets to rotate through so that a set is long gone from the 256 MiB
    last-level cache when its turn comes again (> 2 x the cache in flight)."""
    return max(2, -(-2 * MALL_BYTES // (16 * n * T)) + 1)


def scan_bench(args):
    """``--config c4scan`` / ``c3scan``: the scan kernel alone, BASELINE.json's
    second metric ("GAE-scan HBM GB/s").  ``value`` and ``roofline`` are measured
    with the launches rotating through enough distinct buffer sets that inputs
    and outputs really travel to / from HBM; the same launch on buffers that stay
    in the 256 MiB Infinity Cache (what a scan inside a PPO iteration sees for
    its just-written baselines) is reported beside it as ``llc_resident`` --
    cache bandwidth, no fraction of the HBM peak attached."""
    cfg = SCAN_CONFIGS[args.config]
    n, T = cfg['n_rows'], cfg['T']
    torch.cuda.set_device(0)
    sets = scan_sets_for_hbm(n, T)
    t0 = time.perf_counter()
    gbs, us, nbytes = scan_launches(n, T, args.steps, sets, args.warmup)
    elapsed = time.perf_counter() - t0
    gbs_llc, us_llc, _ = scan_launches(n, T, args.steps, 1, args.warmup)
    print(json.dumps({
        'metric': 'GAE-scan HBM GB/s', 'value': gbs, 'unit': 'GB/s',
        'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': us * 1e-3, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32 in memory, f64 recurrences',
        'data': 'synthetic',
        'config': {'workload': '{}: {} rows x {} steps, gamma {} lambda {}, '
                   '16 B per step (r, V in; A, G out), {} buffer sets in '
                   'rotation ({:.0f} MB > 256 MiB Infinity Cache)'.format(
                       cfg['name'], n, T, HYPER['discount'],
                       HYPER['gae_lambda'], sets, sets * nbytes / 1e6),
                   'config_id': args.config, 'parallelism': 'dp1'},
        'roofline': {'kernel': 'gae_scan_rows_kernel<1>', 'bound': 'hbm',
                     'achieved': gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                     'frac': gbs / PEAK_HBM_GBS, 'traffic': None,
                     'frac_of_measured_copy_6290': gbs / 6290.0,
                     'avg_launch_us': us, 'bytes_per_launch': nbytes,
                     'measured': 'HIP events attached to each of the {} '
                                 'launches (ms_per_step is their mean); the '
                                 'whole leg incl. allocation took {:.2f} s of '
                                 'host time'.format(args.steps, elapsed)},
        'llc_resident': {'achieved': gbs_llc, 'unit': 'GB/s',
                         'avg_launch_us': us_llc,
                         'note': 'same launch, same buffers back to back: '
                                 'Infinity-Cache bandwidth below 256 MiB, not '
                                 'HBM'},
    }))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--config', default='c3',
                    choices=sorted(CONFIGS) + sorted(SCAN_CONFIGS))
    ap.add_argument('--cpu-envs', type=int, default=1024,
                    help='envs of the bounded CPU-baseline sample (0: skip)')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-scan-c4', action='store_true',
                    help='skip the C4-footprint scan launches after the timed '
                    'region (roofline_gae_scan_c4)')
    ap.add_argument('--fuse-head', action='store_true',
                    help='compute the head layer inside the loss kernel '
                    '(opt-in; measured neutral at C3)')
    ap.add_argument('--algo', default='ppo', choices=['ppo', 'trpo'],
                    help='trpo: the section-8f.1 widening (conjugate-gradient '
                    'policy step), reported under its own metric name')
    ap.add_argument('--no-head-dgrad-fusion', action='store_true',
                    help='A/B switch: separate launch for the data gradient '
                    'below the head layer')
    ap.add_argument('--no-small-step', action='store_true',
                    help='A/B switch: minibatches of <= 64 rows take the '
                    'per-layer launches instead of the one-launch step')
    ap.add_argument('--one-launch-losses', action='store_true',
                    help='A/B switch: multi-block losses finish through a '
                    'last-ticket block instead of a finalize launch')
    ap.add_argument('--head-forward-fusion', type=int, default=None,
                    choices=[0, 1, 2],
                    help='A/B switch (ga_set_fused_head_forward): 0 separate '
                    'head launch, 1 default, 2 also 256-wide hidden layers')
    ap.add_argument('--dp', action='store_true',
                    help='with --gpus 1: run the DATA-PARALLEL code path on one GPU '
                    '-- a real nccl (= RCCL) process group of one rank, the '
                    'library-owned communicators, reduce -> ncclAllReduce -> Adam '
                    'per optimizer step -- to price its launch chain against the '
                    'plain single-process one')
    ap.add_argument('--no-overlap', action='store_true',
                    help='run the policy and value-function passes one after '
                    'the other on one stream (isolated per-kernel timings)')
    ap.add_argument('--no-split-variant', action='store_true',
                    help='skip the extra timed steps with the opt-in split-operand '
                    '(3 x bf16) k-loops that fill value_split_bf16')
    args = ap.parse_args()

    if args.config in SCAN_CONFIGS:
        return scan_bench(args)
    from garage_amd.distributed import gradient_exchange, init_from_env
    comm = init_from_env()
    if args.dp and comm is None:
        # a world of ONE rank is still a real process group: shard_algo creates the
        # RCCL communicators and every optimizer step goes through the C++ loop's
        # data-parallel branch (a one-rank sum is the identity)
        import socket

        import torch.distributed as dist

        from garage_amd.distributed import Comm
        if args.gpus != 1:
            raise SystemExit('bench.py: --dp prices the data-parallel path on ONE '
                             'GPU; with --gpus N > 1 it is what runs anyway')
        with socket.socket() as sock:
            sock.bind(('127.0.0.1', 0))
            port = sock.getsockname()[1]
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(port))
        torch.cuda.set_device(0)
        dist.init_process_group(backend='nccl', rank=0, world_size=1)
        comm = Comm()
    world = comm.world_size if comm is not None else 1
    rank = comm.rank if comm is not None else 0
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus {} but WORLD_SIZE is {}'.format(
            args.gpus, world))
    cfg = CONFIGS[args.config]
    algo, sampler, pol, S = build_engine(cfg, comm, algo_name=args.algo)
    algo.overlap_updates = not args.no_overlap
    algo.fuse_head = bool(args.fuse_head)
    if args.no_head_dgrad_fusion:
        from garage_amd import _lib
        _lib.load().ga_set_fused_head_dgrad(0)
    if args.no_small_step:
        from garage_amd import _lib
        _lib.load().ga_set_small_step(0)
    if args.one_launch_losses:
        from garage_amd import _lib
        _lib.load().ga_set_one_launch_losses(1)
    if args.head_forward_fusion is not None:
        from garage_amd import _lib
        _lib.load().ga_set_fused_head_forward(args.head_forward_fusion)

    def sync():
        if comm is not None:
            comm.barrier()
        torch.cuda.synchronize()

    itr = 0
    for _ in range(args.warmup):
        one_iteration(algo, sampler, pol, S, itr)
        itr += 1
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_iteration(algo, sampler, pol, S, itr)
        itr += 1
    sync()
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64,
                           device='cuda')
    if comm is not None:
        comm.all_reduce(elapsed, 'max')
    elapsed = float(elapsed.item())

    # Instrumented iterations after the timed region (every GEMM / scan /
    # streaming launch bracketed by HIP events on its stream).  `rows`: policy and
    # value passes on ONE stream, so a launch's duration is the kernel's own;
    # `rows_ovl`: the timed region's mode when that is the overlapped one (there
    # a duration includes the time the kernel shares the chip with the other
    # chain's kernels).
    rows = rows_ovl = None
    if not args.no_roofline:
        algo.overlap_updates = False
        rows = roofline_pass(algo, sampler, pol, S, itr)
        itr += 1
        if not args.no_overlap:
            algo.overlap_updates = True
            rows_ovl = roofline_pass(algo, sampler, pol, S, itr)
    # The opt-in experiment (include/garage_amd.h: ga_set_split_bf16), reported under
    # its own keys: the SAME iteration with the k-loops of the update kernels on
    # v_mfma_f32_32x32x16_bf16 (every fp32 operand as three bf16 terms, six products,
    # fp32 accumulation).  The headline `value` above is exact fp32 and stays so.
    split = None
    # (one GPU only: the N > 1 runs are the driver's scaling measurement of the exact path)
    if (args.config in ('c3', 'c5') and args.algo == 'ppo' and not args.no_split_variant
            and world == 1 and not os.environ.get('GARAGE_AMD_SPLIT_BF16')):
        from garage_amd import _lib
        lib = _lib.load()
        lib.ga_set_split_bf16(1)
        try:
            # (a second engine: the number of split-K slabs is chosen when the
            # workspaces are built, and differs with the faster kernel)
            algo, sampler, pol, S = build_engine(cfg, comm, algo_name=args.algo)
            algo.fuse_head = bool(args.fuse_head)
            algo.overlap_updates = not args.no_overlap
            one_iteration(algo, sampler, pol, S, itr)
            itr += 1
            sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                one_iteration(algo, sampler, pol, S, itr)
                itr += 1
            sync()
            el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64,
                              device='cuda')
            if comm is not None:
                comm.all_reduce(el, 'max')
            split = {'elapsed': float(el.item()), 'rows': None}
            if not args.no_roofline:
                algo.overlap_updates = False
                split['rows'] = roofline_pass(algo, sampler, pol, S, itr)
                itr += 1
        finally:
            lib.ga_set_split_bf16(0)
    if rank != 0:
        return
    ms_per_step = elapsed / args.steps * 1e3
    value = S * world * args.steps / elapsed
    line = {
        'metric': 'env-steps/sec (whole node) {} {} envs'.format(
            args.algo.upper(), cfg['n_envs']),
        'value': value,
        'unit': 'env-steps/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': ms_per_step,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f32',
        'data': 'synthetic',
        'config': {
            'workload': ('{}: obs {} act {}, {} envs/GPU x T={}, '
                         'MLP{} policy + value, {} E={} x {} minibatches, '
                         'gamma {} lambda {} clip {} Adam lr {}, device '
                         'minibatch permutation, agent_infos (mean, log_std) '
                         'stored, policy/value passes {}').format(
                            cfg['name'], cfg['obs_dim'], cfg['act_dim'],
                            cfg['n_envs'], cfg['T'], cfg['hidden'],
                            'PPO' if args.algo == 'ppo' else
                            'TRPO (CG policy step) + value',
                            HYPER['epochs'], n_minibatches(cfg),
                            HYPER['discount'], HYPER['gae_lambda'],
                            HYPER['lr_clip_range'], HYPER['lr'],
                            'serial' if args.no_overlap else
                            'overlapped on 2 streams'),
            'config_id': args.config,
            'parallelism': 'dp{}'.format(world),
        },
    }
    exchange, rccl_ranks = gradient_exchange(algo)
    line['grad_allreduce'] = exchange
    line['rccl_ranks'] = rccl_ranks
    if args.dp:
        line['config']['parallelism'] = 'dp1 (data-parallel code path, one rank)'
    if rows is not None:
        gemms = [rows[k] for k in GEMM_KINDS if rows[k]['launches'] > 0]
        dom = max(gemms, key=lambda r: r['total_ms'])
        tflops = dom['work'] / (dom['total_ms'] * 1e-3) / 1e12
        line['roofline'] = {
            'kernel': dom['kernel'], 'bound': 'mfma', 'achieved': tflops,
            'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
            'frac': tflops / PEAK_FP32_MFMA_TFLOPS, 'traffic': None,
            'launches_per_iteration': dom['launches'],
            'avg_launch_us': dom['total_ms'] * 1e3 / dom['launches'],
            'measured': 'HIP events attached to every launch of the kernel in one '
                        'instrumented iteration after the timed region, policy '
                        'and value passes on one stream (the kernel alone on the '
                        'chip; agrees with profiles/*no_overlap_kernel_stats.csv)',
        }
        if rows_ovl is not None:
            o = rows_ovl[rows.index(dom)]
            otf = o['work'] / (o['total_ms'] * 1e-3) / 1e12
            line['roofline_overlapped'] = {
                'kernel': o['kernel'], 'achieved': otf, 'unit': 'TFLOP/s',
                'frac': otf / PEAK_FP32_MFMA_TFLOPS,
                'avg_launch_us': o['total_ms'] * 1e3 / o['launches'],
                'note': 'the same kernel during an instrumented iteration in the '
                        'timed region\'s mode (2 streams): durations include '
                        'sharing the chip with the other chain; agrees with '
                        'profiles/*_overlap_kernel_stats.csv',
            }
        traffic = load_traffic()
        key = dom['kernel'].split(' (')[0].replace(',', ', ')
        if args.config == 'c3' and key in traffic:
            line['roofline']['traffic'] = traffic[key]['hbm_bytes']
            line['roofline']['traffic_source'] = TRAFFIC_FILE
        # all GEMM launches of the iteration against its wall time: with the two
        # update chains overlapped, per-kernel durations include time sharing,
        # so this aggregate is the utilisation figure that adds up
        all_flops = sum(rows[k]['work'] for k in GEMM_KINDS)
        agg = all_flops / (ms_per_step * 1e-3) / 1e12
        line['mfma_aggregate'] = {
            'achieved': agg, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
            'frac': agg / PEAK_FP32_MFMA_TFLOPS,
            'flops_per_iteration': all_flops,
            'note': 'algorithmic GEMM flops of one iteration / ms_per_step '
                    '(rollout, scan, losses and optimiser time included)',
        }
        scan = rows[6]
        if scan['launches'] > 0:
            gbs = scan['work'] / (scan['total_ms'] * 1e-3) / 1e9
            line['roofline_gae_scan'] = {
                'kernel': scan['kernel'], 'bound': 'hbm', 'achieved': gbs,
                'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                'frac': gbs / PEAK_HBM_GBS, 'traffic': None,
                'frac_of_measured_copy_6290': gbs / 6290.0,
                'avg_launch_us': scan['total_ms'] * 1e3 / scan['launches'],
                'bytes_per_launch': scan['work'] / scan['launches'],
            }
            for name in ('gae_scan_rows_kernel<1>', 'gae_scan_rows_kernel',
                         'gae_scan_kernel<true>'):
                if args.config == 'c3' and name in traffic:
                    line['roofline_gae_scan']['traffic'] = \
                        traffic[name]['hbm_bytes']
                    line['roofline_gae_scan']['traffic_source'] = TRAFFIC_FILE
                    break
        line['kernels'] = [
            dict(kernel=r['kernel'], launches=r['launches'],
                 total_ms=round(r['total_ms'], 3),
                 avg_us=round(r['total_ms'] * 1e3 / max(1, r['launches']), 2))
            for r in rows if r['launches'] > 0
        ]
        if world == 1 and args.config == 'c3' and not args.no_scan_c4:
            # BASELINE.json's second metric at the footprint north_star's >= 40 %
            # target is meant for (C4-sized buffers), measured
            # after the timed region: 20 launches over 32768 x 256, rotating
            # through buffer sets so that every launch streams from / to HBM
            n4, T4 = SCAN_CONFIGS['c4scan']['n_rows'], SCAN_CONFIGS['c4scan']['T']
            sets = scan_sets_for_hbm(n4, T4)
            gbs4, us4, nb4 = scan_launches(n4, T4, 20, sets)
            line['roofline_gae_scan_c4'] = {
                'kernel': 'gae_scan_rows_kernel<1>', 'bound': 'hbm',
                'achieved': gbs4, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                'frac': gbs4 / PEAK_HBM_GBS,
                'frac_of_measured_copy_6290': gbs4 / 6290.0,
                'avg_launch_us': us4, 'bytes_per_launch': nb4,
                'rows': n4, 'T': T4, 'buffer_sets': sets,
                'note': '20 launches after the timed region at the C4 batch '
                        'footprint (8 x 4096 envs x T = 256), rotating through '
                        '{} buffer sets = {:.0f} MB (> 2 x the 256 MiB Infinity '
                        'Cache): inputs from HBM, outputs to HBM'.format(
                            sets, sets * nb4 / 1e6),
            }
    if os.environ.get('GARAGE_AMD_SPLIT_BF16') == '1':
        line['dtype'] = ('f32 in memory; update-kernel operands as 3 x bf16 (split), '
                         'fp32 accumulate (GARAGE_AMD_SPLIT_BF16=1: opt-in experiment)')
    if split is not None:
        line['value_split_bf16'] = S * world * args.steps / split['elapsed']
        line['ms_per_step_split_bf16'] = split['elapsed'] / args.steps * 1e3
        line['dtype_split_bf16'] = (
            'f32 in memory; operands of the update kernels / wide-layer GEMMs (and of '
            'the evaluation forward) split into 3 bf16 terms each, 6 products on '
            'v_mfma_f32_32x32x16_bf16, fp32 accumulation -- opt-in experiment '
            '(ga_set_split_bf16), gradients as close to fp64 as the exact kernels\' '
            '(profiles/r03_split_error_histogram.json); `value` is exact fp32')
        if split['rows'] is not None:
            line['kernels_split_bf16'] = [
                dict(kernel=r['kernel'], launches=r['launches'],
                     total_ms=round(r['total_ms'], 3),
                     avg_us=round(r['total_ms'] * 1e3 / max(1, r['launches']), 2))
                for r in split['rows'] if r['launches'] > 0
            ]
    if args.cpu_envs > 0 and world == 1 and args.algo == 'ppo':
        line['cpu_baseline'] = cpu_baseline(cfg, min(args.cpu_envs,
                                                     cfg['n_envs']))
    print(json.dumps(line))


if __name__ == '__main__':
    main()
