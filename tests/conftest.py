import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "rdo-ptq_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _usable_cores():
    """Cores this process may use: affinity mask capped by the cgroup CPU quota (the GPU boxes expose 256 hardware threads to a
    16-CPU cgroup: torch's default of one thread per visible CPU makes the CPU oracle crawl there)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per))))
    except Exception:
        pass
    return n


def pytest_configure(config):
    try:
        import torch
        torch.set_num_threads(_usable_cores())
    except Exception:
        pass
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
