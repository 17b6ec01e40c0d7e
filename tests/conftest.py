import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_sde():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "irsde_golden.npz"))


@pytest.fixture(scope="session")
def golden_sde2():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "irsde_golden2.npz"))
