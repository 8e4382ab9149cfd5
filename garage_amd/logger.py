"""Tabular recorder with the ``dowel.tabular`` calls the hot path makes.

garage logs through ``dowel`` (``torch/algos/vpg.py:186-199``,
``_functions.py:262-273``).  When dowel is importable its global ``tabular`` is
used; otherwise this in-memory recorder stands in so the same keys are
available to callers (``tabular.as_dict``).
"""
import contextlib


class Tabular:

    def __init__(self):
        self._values = {}
        self._prefix = ''

    def record(self, key, val):
        self._values[self._prefix + str(key)] = val

    @contextlib.contextmanager
    def prefix(self, prefix):
        old = self._prefix
        self._prefix = old + prefix
        try:
            yield
        finally:
            self._prefix = old

    @property
    def as_dict(self):
        return dict(self._values)

    def clear(self):
        self._values.clear()


messages = []  # text lines of the stand-in `log` (most recent last)


def _log(msg):
    messages.append(str(msg))
    del messages[:-100]


try:  # pragma: no cover - dowel is not installed in the build image
    from dowel import tabular  # noqa: F401
    from dowel import logger as _dowel_logger
    log = _dowel_logger.log
except ImportError:
    tabular = Tabular()
    log = _log
