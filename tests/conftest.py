"""Shared pytest configuration.

Markers:
  gpu -- needs a real MI355X (run by the driver with ``-m gpu`` on the GPU box).
  ref -- needs the reference checkout at /root/reference (build container only;
         auto-skipped elsewhere).
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs an MI355X GPU')
    config.addinivalue_line(
        'markers', 'ref: test imports the reference from /root/reference')


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir('/root/reference/src/garage')
    skip_ref = pytest.mark.skip(reason='/root/reference not present')
    for item in items:
        if 'ref' in item.keywords and not have_ref:
            item.add_marker(skip_ref)


@pytest.fixture(scope='session')
def golden():
    """Loader for the committed golden vectors (``allow_pickle`` stays off)."""

    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'))

    return load
