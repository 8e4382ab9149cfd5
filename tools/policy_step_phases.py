"""Phase timestamps of workgroup 0 of the fused rollout step (policy_fused.hip)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from garage_amd import _lib
lib = _lib.load()
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'c3']
algo, sampler, pol, S = bench.build_engine(cfg, None)
eps = sampler.obtain_samples(0, S, None)
buf = (C.c_longlong * 32)()
assert lib.ga_policy_step_debug(buf) == 1
eps = sampler.obtain_samples(1, S, None)
assert lib.ga_policy_step_debug(buf) == 0
t = np.array(list(buf), dtype=np.int64)
L = len(pol.net.dims) - 1
marks = [(0, 'start'), (1, 'observations staged')] + [
    (2 + l, 'hidden layer %d' % l) for l in range(L - 1)] + [
    (10, 'output layer'), (11, 'sampling + env step')]
for base, what in ((0, 'one step per launch (weights streamed from L2)'),
                   (16, 'middle step of a whole-rollout launch (weights resident)')):
    if t[base + 11] == 0:
        continue
    print(what)
    prev = t[base]
    for i, name in marks[1:]:
        print('  %-24s %6.2f us' % (name, (t[base + i] - prev) / 100.0))
        prev = t[base + i]
    print('  total                    %6.2f us' % ((t[base + 11] - t[base]) / 100.0))
