"""A/B of the small-M GEMM dispatch (64x64 tiles, 128-deep k-steps)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from garage_amd import _lib
from garage_amd.engine import FlatMLP, pad_rows
from tools.microbench import timeit

lib = _lib.load()
dev = torch.device('cuda')
for hs in ((256, 256), (64, 64), (512, 512, 512)):
    O, A = (376, 17) if len(hs) == 3 else (17, 6)
    net = FlatMLP(O, A, hs, dev)
    net.params.normal_(0, 0.1)
    for M in (64, 256, 1024, 4096, 8192, 16384, 32768):
        X = pad_rows(torch.randn(M, O))
        idx = torch.randperm(M, device=dev).to(torch.int32)
        row = []
        for on in (0, 1):
            lib.ga_set_small_m_gemm(on)
            net.forward(X, M, row_idx=idx)
            us_f = timeit(lambda: net.forward(X, M, row_idx=idx))
            dout = net.dout_view(M)
            dout.normal_()
            us_b = timeit(lambda: net.backward(X, M, dout, row_idx=idx))
            row.append((us_f, us_b))
        print('%s M=%5d  fwd %6.1f -> %6.1f us   bwd %6.1f -> %6.1f us' %
              (hs, M, row[0][0], row[1][0], row[0][1], row[1][1]))
lib.ga_set_small_m_gemm(1)
