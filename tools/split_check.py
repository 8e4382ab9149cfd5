"""Developer tool: one training iteration with the split-operand (3 x bf16) k-loops on
against the exact fp32 default -- parameter / Adam-state / scalar differences.
    python tools/split_check.py [case ...]   (cases of tests/test_fused_train_gpu.py)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from garage_amd import _lib  # noqa: E402
import test_fused_train_gpu as T  # noqa: E402

lib = _lib.load()
for case in (sys.argv[1:] or ['c3_shape', 'pipelined_obs18_ragged']):
    spec, batch = T._problem(case)
    opt = (torch.optim.Adam, dict(lr=1e-3))
    res = []
    try:
        for on in (0, 1):
            lib.ga_set_split_bf16(on)
            algo, pol, vf = T._algo(case, spec, opt, epochs=2)
            np.random.seed(11)
            algo._train_once(0, batch)
            res.append((pol.net.params.clone(), vf.net.params.clone(),
                        dict(algo.last_tabular)))
    finally:
        lib.ga_set_split_bf16(0)
    dp = (res[0][0] - res[1][0]).abs().max().item()
    dv = (res[0][1] - res[1][1]).abs().max().item()
    print(case, 'max |d policy params| %.3e  max |d vf params| %.3e' % (dp, dv))
    for k in res[0][2]:
        a, b = res[0][2][k], res[1][2][k]
        print('   %-28s %+.9e %+.9e  d %.2e' % (k, a, b, abs(a - b)))
