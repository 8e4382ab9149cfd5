"""Developer tool: the packed-fp32 / bf16-MFMA hazard seen from the library
(tools/mfma_valu_hazard.hip is the 100-line reproducer).  One
forward + backward pass of a C3-shaped policy network over 1 M rows on the main stream,
repeated, while a second stream runs a register-only MFMA loop (``ga_debug_mfma_burn``):
nothing, fp32 MFMAs, bf16 MFMAs.  Every output must be the same bits every time; with a
library built WITH hipcc's SLP vectorizer (``make EXTRA=-fslp-vectorize``) the
first-layer weight-gradient slabs (skinny_wgrad_kernel<5>: ``v_pk_fma_f32 ...
op_sel:[0,1,0]``) change in ~60 000 of 655 360 elements under the bf16 loop and only
there.

    python tools/hazard_repro_backward.py
    GA_VARIANT_LIB=garage_amd/_C/variants/lib_slp.so python tools/hazard_repro_backward.py
"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bench
from garage_amd import _lib
if os.environ.get('GA_VARIANT_LIB'):  # make slp-variant
    _lib.LIB_PATH = os.path.abspath(os.environ['GA_VARIANT_LIB'])
lib = _lib.load()
cfg = bench.CONFIGS['c3']
algo, sampler, pol, S = bench.build_engine(cfg, None, seed=2, algo_name='trpo')
eps = sampler.obtain_samples(0, S, None)
net = pol.net
M = S
X = torch.randn(M, 17, device='cuda')
Xp = torch.zeros(M, 20, device='cuda'); Xp[:, :17] = X
head = net.forward(Xp, M, keep_acts=True)
d = net.dout_view(M); d.copy_(torch.randn_like(d) * 0.01)
def grad():
    h = net.forward(Xp, M, keep_acts=True).clone()
    a = net._acts[:2 * M * 256].clone()
    net.backward(Xp, M, d)
    return net._slabs.clone(), h, a, net._dacts[:2 * M * 256].clone()
ref = grad()
torch.cuda.synchronize()
side = torch.cuda.Stream()
sink = torch.zeros(16, device='cuda')
lib.ga_debug_mfma_burn.restype = C.c_int
for mode, name in ((0, 'quiet'), (2, 'fp32 MFMA burn'), (1, 'bf16 MFMA burn')):
    bad = []
    for rep in range(4):
        if mode:
            lib.ga_debug_mfma_burn(C.c_int(mode), C.c_int(120000), C.c_int(512), C.c_void_p(sink.data_ptr()), C.c_void_p(side.cuda_stream))
        g = grad()
        torch.cuda.synchronize()
        st = net.n_flat
        sl = (g[0] != ref[0]).view(-1, st)
        reg = {'W1': (net.w_off[0], net.b_off[0]), 'b1': (net.b_off[0], net.w_off[1]), 'W2': (net.w_off[1], net.b_off[1]), 'b2': (net.b_off[1], net.w_off[2]), 'Wh': (net.w_off[2], net.b_off[2]), 'bh': (net.b_off[2], st)}
        bad.append({k: int(sl[:, a:b].sum().item()) for k, (a, b) in reg.items()} | {'head': int((g[1] != ref[1]).sum().item()), 'acts': int((g[2] != ref[2]).sum().item()), 'dacts': int((g[3] != ref[3]).sum().item()), 'maxabs': float((g[0] - ref[0]).abs().max().item())})
    print(name)
    for b in bad: print('   ', b)

# detail of the last bf16 repetition
st = net.n_flat
a = g[0].view(-1, st)[:, net.w_off[0]:net.b_off[0]]
b = ref[0].view(-1, st)[:, net.w_off[0]:net.b_off[0]]
ne = (a != b)
print('splits with differences:', int(ne.any(dim=1).sum().item()), 'of', a.shape[0])
cnt = ne.view(ne.shape[0], 256, 20).sum(dim=0)   # [c][j]
print('differences by narrow column j (sum over splits and c):', cnt.sum(dim=0).tolist())
print('differences by wide column c %% 16:', cnt.sum(dim=1).view(16, 16).sum(dim=0).tolist())
print('differences by wide column c // 64 (column block):', cnt.sum(dim=1).view(4, 64).sum(dim=1).tolist())
if not bool(ne.any()):
    print('(no differences: nothing to detail)')
    sys.exit(0)
sp = int(ne.any(dim=1).nonzero()[0].item())
idx = ne[sp].nonzero().flatten()[:12]
for i in idx.tolist():
    print('  split', sp, 'c', i // 20, 'j', i % 20, 'ref %.6f got %.6f' % (b[sp, i].item(), a[sp, i].item()))
