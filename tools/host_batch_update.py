"""PPO._train_once on a HOST EpisodeBatch (what any other garage sampler hands
over: numpy arrays, float64 rewards, object step types) against the same batch
already resident in HBM: the PCIe- and conversion-inclusive update time."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'c3']
algo, sampler, pol, S = bench.build_engine(cfg, None)
eps = sampler.obtain_samples(0, S, None)
algo._train_once(0, eps)
host = eps.to_host()
nbytes = sum(a.nbytes for a in (host.observations, host.actions, host.rewards,
                                host.last_observations))
for name, batch in (('device', eps), ('host', host), ('device', eps),
                    ('host', host)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    algo._train_once(1, batch)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('%s batch: _train_once %.1f ms (%d samples, %.1f MB of host arrays)'
          % (name, dt * 1e3, S, nbytes / 1e6))
