"""Developer tool: device time of ga_gemm_nt_f32 (C = A B^T, plain epilogue) at one
shape, HIP events around 20 launches: python tools/gemm_time.py M N K [lib.so]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from garage_amd import _lib  # noqa: E402

if len(sys.argv) > 4:
    _lib.LIB_PATH = os.path.abspath(sys.argv[4])
from garage_amd._lib import call, dptr, stream_ptr  # noqa: E402

M, N, K = [int(v) for v in sys.argv[1:4]]
dev = torch.device('cuda')
A = torch.randn(M, K, device=dev)
B = torch.randn(N, K, device=dev)
C = torch.empty(M, N, device=dev)
for _ in range(3):
    call('ga_gemm_nt_f32', dptr(A), K, dptr(B), K, dptr(C), N, M, N, K, stream_ptr())
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    call('ga_gemm_nt_f32', dptr(A), K, dptr(B), K, dptr(C), N, M, N, K, stream_ptr())
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
ref = A[:4096] @ B.t()
err = (C[:4096] - ref).abs().max().item()
print('M %d N %d K %d: %.1f us  %.1f TFLOP/s  (max |diff| vs torch on 4096 rows %.2e)' % (
    M, N, K, us, 2.0 * M * N * K / us / 1e6, err))
