"""``garage_amd``'s classes accept the keywords of their reference counterparts,
with the same defaults (garage's plugin surface is a set of
Python classes; a launcher switches to this package by importing the same names
from it).  ``tests/golden/signatures.json`` was written by
``tests/golden/make_golden.py signatures`` from ``inspect.signature`` of the real
classes."""
import inspect
import json
import os

import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

# reference name -> (module, attribute) here
COUNTERPARTS = {
    'LocalSampler': ('garage_amd.sampler', 'GpuVecSampler'),
    'LocalSampler.from_worker_factory': ('garage_amd.sampler',
                                         'GpuVecSampler.from_worker_factory'),
    'LocalSampler.obtain_samples': ('garage_amd.sampler',
                                    'GpuVecSampler.obtain_samples'),
    'LocalSampler.obtain_exact_episodes':
    ('garage_amd.sampler', 'GpuVecSampler.obtain_exact_episodes'),
    'WorkerFactory': ('garage_amd.sampler', 'WorkerFactory'),
    'VecWorker': ('garage_amd.sampler', 'GpuVecWorker'),
    'FragmentWorker': ('garage_amd.sampler', 'GpuFragmentWorker'),
    'VPG': ('garage_amd.algos', 'VPG'),
    'PPO': ('garage_amd.algos', 'PPO'),
    'TRPO': ('garage_amd.algos', 'TRPO'),
    'GaussianMLPPolicy': ('garage_amd.policies', 'GaussianMLPPolicy'),
    'GaussianMLPValueFunction': ('garage_amd.policies',
                                 'GaussianMLPValueFunction'),
    'OptimizerWrapper': ('garage_amd.optimizers', 'OptimizerWrapper'),
    'ConjugateGradientOptimizer': ('garage_amd.optimizers',
                                   'ConjugateGradientOptimizer'),
    'NormalizedEnv': ('garage_amd.envs', 'NormalizedVecEnv'),
    'EpisodeBatch': ('garage_amd._dtypes', 'EpisodeBatch'),
    'EnvSpec': ('garage_amd._dtypes', 'EnvSpec'),
    'NewEnvUpdate': ('garage_amd.sampler', 'NewEnvUpdate'),
    'SetTaskUpdate': ('garage_amd.sampler', 'SetTaskUpdate'),
    'ExistingEnvUpdate': ('garage_amd.sampler', 'ExistingEnvUpdate'),
    'discount_cumsum': ('garage_amd.functions', 'discount_cumsum'),
    'pad_batch_array': ('garage_amd.functions', 'pad_batch_array'),
    'compute_advantages': ('garage_amd.functions', 'compute_advantages'),
    'filter_valids': ('garage_amd.functions', 'filter_valids'),
    'log_performance': ('garage_amd.functions', 'log_performance'),
    'log_multitask_performance': ('garage_amd.functions',
                                  'log_multitask_performance'),
}
# Defaults that differ on purpose (each one documented where it is defined):
# the vectorised worker is this package's default worker, with one worker per
# sampler (the GPU batch replaces garage's process-level parallelism)
DIFFERENT_DEFAULTS = {
    ('LocalSampler', 'n_workers'), ('LocalSampler', 'worker_class'),
    ('WorkerFactory', 'n_workers'), ('WorkerFactory', 'worker_class'),
    # ConjugateGradientOptimizer's `params` is torch's parameter list; the
    # settings object here ignores it
    ('ConjugateGradientOptimizer', 'params'),
    ('ConjugateGradientOptimizer', 'max_constraint_value'),
}


def _default(v):
    if v is inspect.Parameter.empty:
        return '<required>'
    if v is None or isinstance(v, (bool, int, float, str)):
        return v
    if isinstance(v, (tuple, list)):
        return [_default(x) for x in v]
    if callable(v):
        return '<callable {}>'.format(getattr(v, '__name__', type(v).__name__))
    return '<{}>'.format(type(v).__name__)


def _resolve(module, attr):
    import importlib
    obj = importlib.import_module(module)
    for part in attr.split('.'):
        obj = getattr(obj, part)
    return obj


with open(os.path.join(GOLDEN, 'signatures.json')) as _f:
    REFERENCE = json.load(_f)


@pytest.mark.parametrize('name', sorted(COUNTERPARTS))
def test_same_keywords_and_defaults_as_the_reference(name):
    obj = _resolve(*COUNTERPARTS[name])
    fn = obj.__init__ if inspect.isclass(obj) else obj
    mine = {p.name: p for p in inspect.signature(fn).parameters.values()
            if p.name not in ('self', 'cls')}
    var_kw = any(p.kind is inspect.Parameter.VAR_KEYWORD
                 for p in mine.values())
    order = [p for p in mine if mine[p].kind in (
        inspect.Parameter.POSITIONAL_ONLY,
        inspect.Parameter.POSITIONAL_OR_KEYWORD)]
    ref_order = [p['name'] for p in REFERENCE[name]
                 if p['kind'] == 'POSITIONAL_OR_KEYWORD']
    # positional arguments come in the reference's order (extras go after them)
    assert order[:len(ref_order)] == ref_order, (name, order, ref_order)
    for p in REFERENCE[name]:
        if p['kind'] in ('VAR_KEYWORD', 'VAR_POSITIONAL'):
            continue
        assert p['name'] in mine or var_kw, (name, p['name'])
        if p['name'] not in mine or (name, p['name']) in DIFFERENT_DEFAULTS:
            continue
        assert _default(mine[p['name']].default) == p['default'], (
            name, p['name'], _default(mine[p['name']].default), p['default'])


def test_every_recorded_signature_has_a_counterpart():
    # DefaultWorker: one env per worker is what the vectorised worker replaces;
    # the oracle restates it for the observation-parity tests
    assert set(REFERENCE) - set(COUNTERPARTS) == {'DefaultWorker'}
