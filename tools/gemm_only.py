"""One GEMM shape, many launches (for rocprofv3 --pmc runs)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from garage_amd._lib import call, dptr, stream_ptr
M, N, K = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (32768, 256, 256))]
dev = torch.device('cuda')
A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev)
C = torch.empty(M, N, device=dev)
for _ in range(20):
    call('ga_gemm_nt_f32', dptr(A), K, dptr(B), K, dptr(C), N, M, N, K, stream_ptr())
torch.cuda.synchronize()
