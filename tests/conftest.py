import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _usable_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota.  A GPU box shows every host core to
    os.cpu_count() but grants a share; torch's default thread count then oversubscribes it and the CPU oracle crawls."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    try:
        import torch
        torch.set_num_threads(_usable_cores())
    except ImportError:
        pass


@pytest.fixture(scope="session")
def golden_sde():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "irsde_golden.npz"))


@pytest.fixture(scope="session")
def golden_sde2():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "irsde_golden2.npz"))
