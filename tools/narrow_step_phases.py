"""Phase timestamps of workgroup 0 of the one-launch narrow step (narrow_step.hip)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from garage_amd import _lib
lib = _lib.load()
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'c2']
algo, sampler, pol, S = bench.build_engine(cfg, None)
algo.overlap_updates = False
eps = sampler.obtain_samples(0, S, None)
algo._train_once(0, eps)
buf = (C.c_longlong * 16)()
assert lib.ga_narrow_step_debug(buf) == 1
algo._train_once(1, eps)
assert lib.ga_narrow_step_debug(buf) == 0
t = np.array(list(buf), dtype=np.int64)[:9]
names = ['P0 stage', 'P1 H1', 'P2 H2', 'P3 head', 'P4 loss rows', 'P5-6 dZ2, head grads',
         'P7-8 dW2, dZ1', 'P9 dW1']
for n, d in zip(names, np.diff(t)):
    print('%-22s %6.2f us' % (n, d / 100.0))
print('total                  %6.2f us' % ((t[-1] - t[0]) / 100.0))
