"""Developer tool: device time of a plain fp32 copy at the GAE scan's C3 footprint
(8.4 MB in + 8.4 MB out) and at larger sizes -- the practical floor the scan's
`roofline_gae_scan` should be read against."""
import torch
dev = torch.device('cuda')
def t(fn, reps=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for n in (2 * 1048576, 16 * 1048576, 128 * 1048576):
    x = torch.randn(n, device=dev); y = torch.empty_like(x)
    us = t(lambda: y.copy_(x))
    print('copy %6.1f MB in + out: %6.1f us per launch (back to back)  %.0f GB/s' % (8 * n / 1e6, us, 8 * n / us / 1e3))
