import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_small():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "small_cases.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def golden_digests():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "digests.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def engine():
    """One engine on cuda:0 for the whole GPU session; fails loudly when the HIP library is missing."""
    from fast_slic_amd import Engine
    e = Engine(0, 4)
    yield e
    e.close()
