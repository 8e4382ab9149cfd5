"""Phase timestamps of the one-launch small-minibatch step (small_step.hip)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from garage_amd import _lib
lib = _lib.load()
cfg = bench.CONFIGS['c3mb64']
algo, sampler, pol, S = bench.build_engine(cfg, None)
algo.overlap_updates = False
eps = sampler.obtain_samples(0, S, None)
algo._train_once(0, eps)
buf = (C.c_longlong * 16)()
assert lib.ga_small_step_debug(buf) == 1
algo._train_once(1, eps)
assert lib.ga_small_step_debug(buf) == 0
t_all = np.array(list(buf), dtype=np.int64)
t = t_all[:11]
names = ['stage X/W2cols', 'A.1 H1', 'A.2 H2 own', 'barrier 1', 'B.1 head',
         'B.2 loss', 'B.3-4 grads', 'barrier 2', 'C.1 dZ1', 'C.2 Adam']
for n, d in zip(names, np.diff(t)):
    print('%-16s %6.2f us' % (n, d / 100.0))
print('total            %6.2f us' % ((t[-1] - t[0]) / 100.0))

print('A.2: store w2r %.2f | mma %.2f | sync %.2f | rest %.2f' % (
    (t_all[11] - t_all[2]) / 100., (t_all[12] - t_all[11]) / 100.,
    (t_all[13] - t_all[12]) / 100., (t_all[3] - t_all[13]) / 100.))
print('C.1: load dZ2 %.2f | mma+sync %.2f | rest %.2f' % (
    (t_all[14] - t_all[8]) / 100., (t_all[15] - t_all[14]) / 100.,
    (t_all[9] - t_all[15]) / 100.))
