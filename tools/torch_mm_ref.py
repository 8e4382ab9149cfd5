"""Developer tool: rocBLAS/hipBLASLt fp32 GEMM times at the update shapes (for
comparison with tools/gemm_sweep.py; not used by the product)."""
import torch
dev = torch.device('cuda')
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for M, N, K in [(32768, 256, 64), (32768, 256, 256), (32768, 256, 1024), (32768, 256, 2048), (131072, 256, 2048)]:
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev)
    C = torch.empty(M, N, device=dev)
    us = t(lambda: torch.mm(A, B.t(), out=C))
    print('torch.mm NT M=%d N=%d K=%d: %.1f us %.1f TF/s' % (M, N, K, us, 2.0*M*N*K/us/1e6), flush=True)
# weight-grad shape: (256 x 32768) @ (32768 x 256)
G = torch.randn(32768, 256, device=dev); H = torch.randn(32768, 256, device=dev)
W = torch.empty(256, 256, device=dev)
us = t(lambda: torch.mm(G.t(), H, out=W))
print('torch.mm TN 256x256x32768: %.1f us %.1f TF/s' % (us, 2.0*256*256*32768/us/1e6))
