import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_v1.npz"))


@pytest.fixture(scope="session")
def weights():
    from vqvdb_amd import synth
    return synth.make_weights(seed=0)


@pytest.fixture(scope="session")
def oracle(weights):
    from vqvdb_amd import synth
    from oracle.oracle import Oracle
    return Oracle(weights, [t[0] for t in synth.TENSORS])


def rel_err(a, b):
    """max |a-b| / max|b|  (tensor-relative)"""
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max() / max(float(np.abs(b).max()), 1e-30))
