"""Phase timestamps and launch skew of dgrad_wgrad0_kernel (fused_train.hip)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from garage_amd import _lib
lib = _lib.load()
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'c3']
algo, sampler, pol, S = bench.build_engine(cfg, None)
algo.overlap_updates = False
eps = sampler.obtain_samples(0, S, None)
algo._train_once(0, eps)
buf = (C.c_longlong * 16)()
assert lib.ga_fused_dgrad_debug(buf, 0, 0) == 1
algo._train_once(1, eps)
assert lib.ga_fused_dgrad_debug(buf, 0, 0) == 0
t = np.array(list(buf), dtype=np.int64)[:5]
names = ['k-loop (dZ2 W2)', 'stage accumulators, X rows', 'dZ1 = . (1 - H1^2), H1 from memory',
         'first-layer grad shares']
for n, d in zip(names, np.diff(t)):
    print('%-36s %6.2f us' % (n, d / 100.0))
print('total                                %6.2f us' % ((t[-1] - t[0]) / 100.0))
n = 512
sk = (C.c_longlong * (3 * n))()
assert lib.ga_fused_dgrad_debug(sk, 1, n) == 0
a = np.array(list(sk), dtype=np.int64).reshape(n, 3) / 100.0
t0 = a[:, 0].min()
for name, col in (('workgroup starts', 0), ('k-loop ends', 1), ('workgroup ends', 2)):
    print('%-18s: min %.2f  median %.2f  max %.2f us' % (
        name, a[:, col].min() - t0, np.median(a[:, col]) - t0, a[:, col].max() - t0))
