"""Kernel micro-benchmarks (developer tool; run through gpurun).

    python tools/microbench.py gemm      # GEMM core at the C3 layer shapes
    python tools/microbench.py scan      # GAE scan at C3 / larger sizes
    python tools/microbench.py mlp       # forward / backward of one minibatch
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from garage_amd._lib import call, dptr, stream_ptr  # noqa: E402
from garage_amd.engine import FlatMLP, gae_scan, pad_rows  # noqa: E402


def timeit(fn, reps=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(
        enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3  # us


def bench_gemm():
    dev = torch.device('cuda')
    for (M, N, K) in [(32768, 256, 256), (32768, 256, 20), (4096, 256, 256),
                      (1048576, 256, 256), (32768, 512, 512)]:
        A = torch.randn(M, K, device=dev)
        B = torch.randn(N, K, device=dev)
        Cm = torch.empty(M, N, device=dev)
        us = timeit(lambda: call('ga_gemm_nt_f32', dptr(A), K, dptr(B), K,
                                 dptr(Cm), N, M, N, K, stream_ptr()))
        print('gemm_nt plain M=%d N=%d K=%d: %.1f us  %.1f TF/s' %
              (M, N, K, us, 2.0 * M * N * K / us / 1e6))


def bench_mlp():
    dev = torch.device('cuda')
    for (O, A, hs, M) in [(17, 6, (256, 256), 32768), (17, 1, (256, 256), 32768),
                          (17, 6, (256, 256), 4096)]:
        net = FlatMLP(O, A, hs, dev)
        net.params.normal_(0, 0.1)
        X = pad_rows(torch.randn(M, O))
        idx = torch.randperm(M, device=dev).to(torch.int32)
        net.forward(X, M)
        us_f = timeit(lambda: net.forward(X, M, row_idx=idx))
        dout = net.dout_view(M)
        dout.normal_()
        us_b = timeit(lambda: net.backward(X, M, dout, row_idx=idx))
        us_r = timeit(lambda: net.reduce_grads())
        us_a = timeit(lambda: net.adam_step(1e-4))
        fl = 2.0 * M * sum(a * b for a, b in zip(net.dims[:-1], net.dims[1:]))
        print('mlp %s M=%d: fwd %.1f us (%.1f TF/s)  bwd %.1f us (%.1f TF/s)'
              '  reduce %.1f us  adam %.1f us' %
              (net.dims, M, us_f, fl / us_f / 1e6, us_b, 2 * fl / us_b / 1e6,
               us_r, us_a))


def bench_scan():
    dev = torch.device('cuda')
    for (n, T) in [(4096, 256), (32768, 256), (8192, 1024), (65536, 128)]:
        r = torch.randn(n, T, device=dev)
        v = torch.randn(n, T, device=dev)
        adv, ret = torch.empty_like(r), torch.empty_like(r)
        us = timeit(lambda: gae_scan(r, v, discount=0.99, gae_lambda=0.97,
                                     max_episode_length=T, adv=adv, ret=ret),
                    reps=100)
        off = torch.arange(0, n * T + 1, T, device=dev, dtype=torch.int64)
        us2 = timeit(lambda: gae_scan(r.view(-1), v.view(-1), discount=0.99,
                                      gae_lambda=0.97, max_episode_length=T,
                                      offsets=off, max_len=T,
                                      adv=adv.view(-1), ret=ret.view(-1)),
                     reps=100)
        gb = 16.0 * n * T / 1e9
        print('gae_scan n=%d T=%d: padded %.1f us (%.0f GB/s)  packed %.1f us '
              '(%.0f GB/s)' % (n, T, us, gb / us * 1e6, us2, gb / us2 * 1e6))


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if what in ('gemm', 'all'):
        bench_gemm()
    if what in ('mlp', 'all'):
        bench_mlp()
    if what in ('scan', 'all'):
        bench_scan()
