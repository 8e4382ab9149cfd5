"""Developer tool: which rocBLAS / hipBLASLt fp32 kernels torch.mm picks at the update
shapes (run under `rocprofv3 --kernel-trace --stats`; the Tensile kernel names carry
the macro tile, MFMA instruction, k depth and LDS options) -- comparison only, not
used by the product."""
import torch

dev = torch.device('cuda')


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = (torch.cuda.Event(enable_timing=True),
            torch.cuda.Event(enable_timing=True))
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for M, N, K in [(32768, 256, 256), (65536, 512, 512), (65536, 512, 376)]:
    X = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev)
    Y = torch.empty(M, N, device=dev)
    us = t(lambda: torch.mm(X, W.t(), out=Y))  # forward: X W^T
    print('fwd  X[%d,%d] W^T[%d,%d]: %.1f us %.1f TF/s' % (
        M, K, K, N, us, 2.0 * M * N * K / us / 1e6), flush=True)
    D = torch.randn(M, N, device=dev)
    G = torch.empty(M, K, device=dev)
    us = t(lambda: torch.mm(D, W, out=G))  # data gradient: dZ W
    print('dgrad D[%d,%d] W[%d,%d]: %.1f us %.1f TF/s' % (
        M, N, N, K, us, 2.0 * M * N * K / us / 1e6), flush=True)
    dW = torch.empty(N, K, device=dev)
    us = t(lambda: torch.mm(D.t(), X, out=dW))  # weight gradient: dZ^T X
    print('wgrad D^T[%d,%d] X[%d,%d]: %.1f us %.1f TF/s' % (
        N, M, M, K, us, 2.0 * M * N * K / us / 1e6), flush=True)
