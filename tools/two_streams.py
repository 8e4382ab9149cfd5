"""Do two independent MLP fwd+bwd chains overlap on two HIP streams?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from garage_amd.engine import FlatMLP, pad_rows

dev = torch.device('cuda')
M = 32768
nets = []
for out in (6, 1):
    net = FlatMLP(17, out, (256, 256), dev)
    net.params.normal_(0, 0.1)
    nets.append(net)
X = pad_rows(torch.randn(M * 4, 17))
idx = [torch.randperm(M * 4, device=dev)[:M].to(torch.int32) for _ in range(2)]
for n in nets:
    n.forward(X, M)
    n.dout_view(M).normal_()

def step(net, i):
    net.forward(X, M, row_idx=idx[i])
    net.backward(X, M, net.dout_view(M), row_idx=idx[i])
    net.reduce_grads()
    net.adam_step(1e-4)

def run(streams, reps=40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for i, net in enumerate(nets):
            with torch.cuda.stream(streams[i]):
                step(net, i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6

s0 = torch.cuda.current_stream()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for _ in range(3):
    run([s0, s0], 5); run([sa, sb], 5)
print('one stream : %.1f us per (policy step + value step)' % run([s0, s0]))
print('two streams: %.1f us per (policy step + value step)' % run([sa, sb]))
