"""garage_amd -- MI355X-native on-policy rollout + PPO update engine.

A drop-in for garage's ``LocalSampler``/``VecWorker`` -> ``discount_cumsum`` /
``compute_advantages`` -> ``PPO._train_once`` path, built
on hand-written gfx950 HIP kernels behind the C ABI in ``include/garage_amd.h``.
"""
__version__ = '0.1.0'
