"""In-container harness that imports the REAL reference (``/root/reference/src``).

Test infrastructure only.  It is used by ``tests/golden/make_golden.py`` to
generate the committed golden vectors and by the optional ``ref``-marked tests
that compare the oracle with the live reference.  Nothing here travels to the
GPU box in a usable form (``/root/reference`` does not exist there) and the
product package never imports it.

The reference pins third-party packages that are absent from this image and
cannot be installed (no network): ``akro``, ``dowel``, ``gym``, ``tensorflow``
(+ a few optional ones).  None of them performs arithmetic on the path we
pin: ``dowel`` is logging, ``gym``/``tensorflow`` are imported at module
import time only, and ``akro`` supplies space *metadata*
(shape / flat_dim / contains).  They are satisfied with in-memory modules
(SURVEY.md Appendix A); all numerics under test come from the real garage,
torch, numpy and scipy code.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types
from unittest import mock

import numpy as np

REF_SRC = '/root/reference/src'

_MOCKED = {
    'dowel', 'gym', 'tensorflow', 'tensorflow_probability', 'ray', 'cma',
    'setproctitle', 'skimage', 'torchvision', 'mujoco_py', 'dm_control',
    'pybullet', 'pybullet_envs', 'metaworld', 'glfw'
}


class _MockFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):

    def find_spec(self, name, path=None, target=None):
        if name.split('.')[0] in _MOCKED:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = mock.MagicMock(name=spec.name)
        m.__path__ = []
        m.__spec__ = spec
        m.__loader__ = self
        m.__name__ = spec.name
        return m

    def exec_module(self, module):
        return None


def _make_akro():
    akro = types.ModuleType('akro')

    class Space:
        pass

    class Box(Space):

        def __init__(self, low, high, shape=None, dtype=np.float32):
            if shape is None:
                low = np.asarray(low, dtype=dtype)
                high = np.asarray(high, dtype=dtype)
                shape = low.shape
            else:
                low = np.full(shape, low, dtype=dtype)
                high = np.full(shape, high, dtype=dtype)
            self.low, self.high, self.shape, self.dtype = (low, high,
                                                           tuple(shape),
                                                           np.dtype(dtype))

        @property
        def flat_dim(self):
            return int(np.prod(self.shape))

        @property
        def bounds(self):
            return self.low, self.high

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape

        def flatten(self, x):
            return np.asarray(x).flatten()

        def unflatten(self, x):
            return np.asarray(x).reshape(self.shape)

        def flatten_n(self, xs):
            xs = np.asarray(xs)
            return xs.reshape((xs.shape[0], -1))

        def unflatten_n(self, xs):
            xs = np.asarray(xs)
            return xs.reshape((xs.shape[0], ) + self.shape)

        def sample(self):
            return np.random.uniform(-1, 1, self.shape).astype(self.dtype)

        def __eq__(self, other):
            return (isinstance(other, Box) and self.shape == other.shape
                    and np.allclose(self.low, other.low)
                    and np.allclose(self.high, other.high))

        def __hash__(self):
            return hash(self.shape)

    class Discrete(Space):

        def __init__(self, n):
            self.n = n
            self.shape = ()
            self.dtype = np.dtype(np.int64)

        @property
        def flat_dim(self):
            return self.n

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == () and 0 <= int(x) < self.n

        def flatten(self, x):
            r = np.zeros(self.n)
            r[int(x)] = 1
            return r

        def unflatten(self, x):
            return int(np.nonzero(x)[0][0])

        def flatten_n(self, xs):
            r = np.zeros((len(xs), self.n))
            r[np.arange(len(xs)), np.asarray(xs, dtype=int)] = 1
            return r

        def sample(self):
            return np.random.randint(self.n)

        def __eq__(self, other):
            return isinstance(other, Discrete) and self.n == other.n

        def __hash__(self):
            return hash(self.n)

    class Dict(Space):
        pass

    class Tuple(Space):
        pass

    class Image(Box):
        pass

    akro.Space, akro.Box, akro.Discrete = Space, Box, Discrete
    akro.Dict, akro.Tuple, akro.Image = Dict, Tuple, Image
    akro.from_gym = lambda s, **kw: s
    return akro


_INSTALLED = False


def available():
    return os.path.isdir(REF_SRC)


def install():
    """Make ``import garage`` resolve to the real reference sources."""
    global _INSTALLED
    if _INSTALLED:
        return
    if not available():
        raise RuntimeError('reference sources not present at ' + REF_SRC)
    sys.dont_write_bytecode = True  # /root/reference must stay pristine
    sys.meta_path.insert(0, _MockFinder())
    sys.modules['akro'] = _make_akro()
    sys.path.insert(0, REF_SRC)
    _INSTALLED = True


class TabularRecorder:
    """Stands in for ``dowel.tabular`` to capture the scalars VPG logs."""

    def __init__(self):
        self.values = {}
        self._prefix = ''

    def record(self, key, val):
        self.values[self._prefix + key] = val

    def prefix(self, p):
        rec = self

        class _Ctx:

            def __enter__(self_inner):
                self_inner.old = rec._prefix
                rec._prefix = rec._prefix + p

            def __exit__(self_inner, *a):
                rec._prefix = self_inner.old

        return _Ctx()
