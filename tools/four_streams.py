"""Row-split sub-chains: do 4 half-size chains on 4 streams beat 2 full-size on 2?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from garage_amd.engine import FlatMLP, pad_rows

dev = torch.device('cuda')
def make(M, out):
    net = FlatMLP(17, out, (256, 256), dev)
    net.params.normal_(0, 0.1)
    net.forward(X, M); net.dout_view(M).normal_()
    return net
Mfull = 32768
X = pad_rows(torch.randn(Mfull * 4, 17))
def idx(M): return torch.randperm(Mfull * 4, device=dev)[:M].to(torch.int32)

def step(net, M, ix):
    net.forward(X, M, row_idx=ix)
    net.backward(X, M, net.dout_view(M), row_idx=ix)

def run(nets, Ms, ixs, streams, reps=40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        for net, M, ix, st in zip(nets, Ms, ixs, streams):
            with torch.cuda.stream(st):
                step(net, M, ix)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6

streams = [torch.cuda.Stream() for _ in range(4)]
full = [make(Mfull, 6), make(Mfull, 1)]
half = [make(Mfull // 2, 6), make(Mfull // 2, 6), make(Mfull // 2, 1), make(Mfull // 2, 1)]
fi = [idx(Mfull), idx(Mfull)]; hi = [idx(Mfull // 2) for _ in range(4)]
for _ in range(2):
    run(full, [Mfull] * 2, fi, streams[:2], 5); run(half, [Mfull // 2] * 4, hi, streams, 5)
print('2 chains x 32768 rows on 2 streams: %.1f us' % run(full, [Mfull] * 2, fi, streams[:2]))
print('4 chains x 16384 rows on 4 streams: %.1f us' % run(half, [Mfull // 2] * 4, hi, streams))
print('2 chains x 32768 rows on 1 stream : %.1f us' % run(full, [Mfull] * 2, fi, [streams[0]] * 2))
