import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # before the HIP runtime initialises: see joligen_amd/__init__.py

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def _usable_cores():
    """host cores this process may actually use: min(affinity mask, cgroup CPU quota).  The CPU oracle legs of the GPU tests run
    torch-CPU kernels; with torch's default (one OpenMP thread per machine core) a quota-limited box oversubscribes its share
    ~10x and the oracle runs ~10x slower than on the cores it really has."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, min(n, 32))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    try:
        import torch

        torch.set_num_threads(_usable_cores())
    except Exception:
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
