"""Developer tool: cProfile of one iteration of a bench config (host-side time)."""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'c5']
algo, sampler, pol, S = bench.build_engine(cfg, None)
for it in range(2):
    eps = sampler.obtain_samples(it, S, None)
    algo._train_once(it, eps)
torch.cuda.synchronize()
for it in range(2, 4):
    t0 = time.perf_counter()
    eps = sampler.obtain_samples(it, S, None)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    algo._train_once(it, eps)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('iteration %d: sample %.1f ms, train %.1f ms' % (it, 1e3 * (t1 - t0), 1e3 * (t2 - t1)))
pr = cProfile.Profile()
pr.enable()
eps = sampler.obtain_samples(4, S, None)
algo._train_once(4, eps)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(30)
