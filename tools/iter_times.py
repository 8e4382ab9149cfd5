"""Developer tool: wall time of each of N consecutive iterations of a bench config
(device-synchronised), and the allocator's reserved / allocated bytes after each:
    python tools/iter_times.py c5 10"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'c5']
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
algo, sampler, pol, S = bench.build_engine(cfg, None)
for it in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eps = sampler.obtain_samples(it, S, None)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    algo._train_once(it, eps)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('iteration %2d: rollout %7.1f ms  update %8.1f ms  samples %d  '
          'allocated %.2f GB reserved %.2f GB' % (
              it, (t1 - t0) * 1e3, (t2 - t1) * 1e3, eps.n_samples,
              torch.cuda.memory_allocated() / 1e9,
              torch.cuda.memory_reserved() / 1e9), flush=True)
