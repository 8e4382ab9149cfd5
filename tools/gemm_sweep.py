"""GEMM core: device time (HIP events attached to the dispatch) over a K / M sweep.

    python tools/gemm_sweep.py

Separates the steady-state k-loop rate from the per-launch prologue/epilogue:
time(K) = t0 + K * slope at fixed M, N.
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from garage_amd import _lib  # noqa: E402
from garage_amd._lib import call, dptr, stream_ptr  # noqa: E402


def timed(fn, reps=20, warm=3, kind=0):
    lib = _lib.load()
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    lib.ga_prof_enable(1)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    lib.ga_prof_enable(0)
    out = (C.c_double * 21)()
    lib.ga_prof_collect(out, 7)
    return out[3 * kind] / max(out[3 * kind + 2], 1) * 1e3  # us


def main():
    dev = torch.device('cuda')
    N = 256
    Ms, Ks = (32768, 65536, 131072), (32, 64, 128, 256, 512, 1024, 2048)
    if len(sys.argv) > 1:  # A/B run of a variant library, reduced shape list
        _lib.LIB_PATH = os.path.abspath(sys.argv[1])
        Ms, Ks = (32768,), (64, 256, 1024, 2048)
        print('lib', sys.argv[1])
    for M in Ms:
        for K in Ks:
            A = torch.randn(M, K, device=dev)
            B = torch.randn(N, K, device=dev)
            Cm = torch.empty(M, N, device=dev)
            us = timed(lambda: call('ga_gemm_nt_f32', dptr(A), K, dptr(B), K,
                                    dptr(Cm), N, M, N, K, stream_ptr()))
            print('M=%6d N=%d K=%4d: %7.1f us  %6.1f TF/s' %
                  (M, N, K, us, 2.0 * M * N * K / us / 1e6), flush=True)


if __name__ == '__main__':
    main()
