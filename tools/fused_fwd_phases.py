"""Phase timestamps of one workgroup of fwd_head_loss_kernel (fused_train.hip)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from garage_amd import _lib
if os.environ.get('GA_VARIANT_LIB'):  # tools/ablate_fused_fwd.sh
    _lib.LIB_PATH = os.path.abspath(os.environ['GA_VARIANT_LIB'])
lib = _lib.load()
cfg = dict(bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'c3'])
if len(sys.argv) > 2:  # minibatches per epoch: 64 -> 256 tiles, one workgroup per CU
    cfg['minibatches'] = int(sys.argv[2])
algo, sampler, pol, S = bench.build_engine(cfg, None)
algo.overlap_updates = False
eps = sampler.obtain_samples(0, S, None)
algo._train_once(0, eps)
buf = (C.c_longlong * 16)()
assert lib.ga_fused_fwd_debug(buf) == 1
algo._train_once(1, eps)
assert lib.ga_fused_fwd_debug(buf) == 0
t = np.array(list(buf), dtype=np.int64)[:9]
names = ['prologue (W1, X, first tiles)', 'k-loop', 'E1 tanh(acc + bias) -> stage',
         'E2 barrier', 'E3 head', 'E4 loss rows', 'E5 dZ store (E6 in the 2nd grid half)',
         'E6 head grad shares']
for n, d in zip(names, np.diff(t)):
    print('%-30s %6.2f us' % (n, d / 100.0))
print('total                          %6.2f us' % ((t[-1] - t[0]) / 100.0))

n = min(512, (S // bench.n_minibatches(cfg) + 63) // 64)
print('workgroups per launch:', n)
sk = (C.c_longlong * (3 * n))()
assert lib.ga_fused_fwd_debug_skew(sk, n) == 0
a = np.array(list(sk), dtype=np.int64).reshape(n, 3) / 100.0
t0 = a[:, 0].min()
print('workgroup starts  : min %.2f  median %.2f  max %.2f us' % (
    0.0, np.median(a[:, 0]) - t0, a[:, 0].max() - t0))
print('k-loop ends       : min %.2f  median %.2f  max %.2f us' % (
    a[:, 1].min() - t0, np.median(a[:, 1]) - t0, a[:, 1].max() - t0))
print('workgroup ends    : min %.2f  median %.2f  max %.2f us' % (
    a[:, 2].min() - t0, np.median(a[:, 2]) - t0, a[:, 2].max() - t0))
d = a[:, 2] - a[:, 0]
print('workgroup duration: min %.2f  median %.2f  max %.2f us' % (d.min(), np.median(d), d.max()))

# developer build (make clean; make EXTRA=-DGA_FT_LOOP_STAMPS): where thread 0 of each
# workgroup spends the k-loop
import ctypes
lib.ga_fused_fwd_debug_loop.restype = ctypes.c_int
lp = (C.c_longlong * (4 * n))()
if lib.ga_fused_fwd_debug_loop(lp, n) == 0:
    b = np.array(list(lp), dtype=np.float64).reshape(n, 4)
    if b.sum() > 0:
        tot = b.sum(axis=1)
        order = np.argsort(a[:, 1])  # by end of k-loop
        for name, sel in (('fastest 64', order[:64]), ('slowest 64', order[-64:]),
                          ('all', order)):
            m = b[sel].mean(axis=0)
            print('%-10s ticks/k-loop: reads+MFMA %8.0f  barrier1 %8.0f  stores %8.0f  '
                  'barrier2 %8.0f  (sum %8.0f)' % (name, m[0], m[1], m[2], m[3], m.sum()))
