import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def kats():
    return dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'reference_kats.npz')))


@pytest.fixture(scope='session')
def lih_walker(kats):
    """The reference's canonical LiH test walker, recovered from the 'ne' edges golden:
    ne[0] = r - R_Li with R_Li = 0."""
    return np.asarray(kats['lih_edges_ne'][0], np.float64)
