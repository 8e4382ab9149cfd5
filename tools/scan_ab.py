"""A/B of the GAE scan fast path (developer tool; run through gpurun):
steps per lane 4 vs 8, isolated launches (HIP events on the dispatch, what
bench.py reports) and back-to-back launches, next to a plain copy of the bytes."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from garage_amd import _lib  # noqa: E402
from garage_amd.engine import gae_scan  # noqa: E402


def isolated(fn, reps=40, before=None):
    lib = _lib.load()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    lib.ga_prof_enable(1)
    for _ in range(reps):
        if before is not None:
            lib.ga_prof_enable(0)
            before()
            torch.cuda.synchronize()
            lib.ga_prof_enable(1)
        fn()
        torch.cuda.synchronize()
    lib.ga_prof_enable(0)
    out = (C.c_double * 33)()
    lib.ga_prof_collect(out, 11)
    return out[18] / max(1.0, out[20]) * 1e3  # us per launch, kind 6


def stream_rate(fn, reps=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(
        enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    lib = _lib.load()
    dev = torch.device('cuda')
    for (n, T) in [(4096, 256), (4096, 128), (32768, 256)]:
        r = torch.randn(n, T, device=dev)
        v = torch.randn(n, T, device=dev)
        adv, ret = torch.empty_like(r), torch.empty_like(r)
        gb = 16.0 * n * T / 1e9

        def run():
            gae_scan(r, v, discount=0.99, gae_lambda=0.97, max_episode_length=T,
                     adv=adv, ret=ret)

        for steps in (4, 8):
            lib.ga_set_gae_rows_steps_per_lane(steps)
            iso, thr = isolated(run), stream_rate(run)
            print('n=%d T=%d steps/lane=%d: isolated %.2f us (%.0f GB/s, %.2f of '
                  '8 TB/s)  back-to-back %.2f us' %
                  (n, T, steps, iso, gb / iso * 1e6, gb / iso * 1e6 / 8000, thr))
        lib.ga_set_gae_rows_steps_per_lane(4)
        # inputs from HBM: a 1 GiB write in between evicts L2 and the Infinity Cache
        big = torch.empty(1 << 28, device=dev)
        cold = isolated(run, reps=15, before=lambda: big.fill_(1.0))

        def half_warm():
            big.fill_(1.0)
            v.mul_(1.0)  # the baselines were just written by the value forward

        half = isolated(run, reps=15, before=half_warm)
        print('   cold inputs: %.2f us (%.2f of 8 TB/s); values warm, rewards '
              'cold: %.2f us (%.2f)' % (cold, gb / cold * 1e6 / 8000, half,
                                        gb / half * 1e6 / 8000))
        del big
        src = torch.cat([r.view(-1), v.view(-1)])
        dst = torch.empty_like(src)
        cp = stream_rate(lambda: dst.copy_(src))
        print('   copy of the same bytes, back-to-back: %.2f us (%.0f GB/s)' %
              (cp, gb / cp * 1e6))


if __name__ == '__main__':
    main()
