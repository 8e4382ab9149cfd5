"""How long does the host take to ENQUEUE one iteration (no device sync)?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
cfg = bench.CONFIGS['c3']
algo, sampler, pol, S = bench.build_engine(cfg, None)
for it in range(2):
    bench.one_iteration(algo, sampler, pol, S, it)
torch.cuda.synchronize()
for it in range(3):
    t0 = time.perf_counter()
    eps = sampler.obtain_samples(it, S, None)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    algo._train_once(it, eps)
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print('rollout: host %.1f ms, +sync %.1f ms | update: host %.1f ms, +sync %.1f ms'
          % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
